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

struct fmac_f64_dpp {
  static __device__ __forceinline__ void run(double& x, double y, double z) { asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y), "v"(z)); }
  static const char* label() { return "fmac_f64_dpp"; }
};
struct fmac_f64 {
  static __device__ __forceinline__ void run(double& x, double y, double z) { asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(x) : "v"(y), "v"(z)); }
  static const char* label() { return "fmac_f64_e32"; }
};
struct mov_b64_dpp {
  static __device__ __forceinline__ void run(double& x, double y, double z) { asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y)); }
  static const char* label() { return "mov_b64_dpp"; }
};
struct mov_b32_dpp {
  static __device__ __forceinline__ void run(double& x, double y, double z) {
    int lo = __double2loint(x);
    asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(lo) : "v"(__double2loint(y)));
    x = __hiloint2double(__double2hiint(x), lo);
  }
  static const char* label() { return "mov_b32_dpp"; }
};
struct readlane_pair_fma {   // the multiplier path the panel used before: two v_readlane_b32 + one v_fma_f64 with an SGPR-pair operand
  static __device__ __forceinline__ void run(double& x, double y, double z) {
    asm volatile("v_readlane_b32 s20, %1, 5\n\tv_readlane_b32 s21, %2, 5\n\tv_fma_f64 %0, %3, s[20:21], %0" : "+v"(x) : "v"(__double2loint(y)), "v"(__double2hiint(y)), "v"(z) : "s20", "s21");
  }
  static const char* label() { return "readlane x2 + fma"; }
};
struct readlane_b32 {
  static __device__ __forceinline__ void run(double& x, double y, double z) {
    asm volatile("v_readlane_b32 s20, %0, 5" : : "v"(__double2loint(x)) : "s20");
  }
  static const char* label() { return "readlane_b32"; }
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

// dependent-issue latency: ONE wavefront, ONE chain (each instruction consumes the previous result)
template <class Op>
__global__ void k_latency(double* out, long long* ticks, int n) {
  double x = 1.0 + threadIdx.x * 1e-3;
  const double y = 1.0000001, z = 1e-9;
  const long long t0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) Op::run(x, y, z);
  }
  const long long t1 = wall_clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
struct dep_fmac_dpp_src {   // the chain runs through the DPP (broadcast) source operand
  static __device__ __forceinline__ void run(double& x, double y, double z) {
    double acc = z;
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y));
    x = acc;
  }
  static const char* label() { return "s_nop 1 + fmac_f64_dpp (dpp src chain)"; }
};
struct dep_mov_dpp {
  static __device__ __forceinline__ void run(double& x, double y, double z) { asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x)); }
  static const char* label() { return "s_nop 1 + mov_b64_dpp"; }
};
struct dep_readlane_fma {   // x -> readlane pair -> fma with the SGPR pair -> x
  static __device__ __forceinline__ void run(double& x, double y, double z) {
    asm volatile("v_readlane_b32 s20, %1, 5\n\tv_readlane_b32 s21, %2, 5\n\tv_fma_f64 %0, %3, s[20:21], %4" : "=v"(x) : "v"(__double2loint(x)), "v"(__double2hiint(x)), "v"(y), "v"(z) : "s20", "s21");
  }
  static const char* label() { return "readlane x2 -> fma (chain)"; }
};
// issue interval of ONE wavefront alone on its SIMD: 8 independent chains
template <class Op>
void MeasureSingleWave(double* out, long long* ticks) {
  const int n = 20000;
  long long t;
  hipLaunchKernelGGL(k_rate<Op>, dim3(1), dim3(64), 0, 0, out, ticks, n);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  printf("single wavefront, independent %-20s %.2f ns per instruction\n", Op::label(), t * 10.0 / (n * 8.0));
}
template <class Op>
void MeasureLatency(double* out, long long* ticks) {
  const int n = 20000;
  long long t;
  hipLaunchKernelGGL(k_latency<Op>, dim3(1), dim3(64), 0, 0, out, ticks, n);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  printf("latency %-40s %.2f ns per dependent instruction\n", Op::label(), t * 10.0 / (n * 8.0));
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
  Measure<fmac_f64>(out, ticks, fma_ns);
  Measure<fmac_f64_dpp>(out, ticks, fma_ns);
  Measure<mov_b64_dpp>(out, ticks, fma_ns);
  Measure<mov_b32_dpp>(out, ticks, fma_ns);
  Measure<readlane_b32>(out, ticks, fma_ns);
  Measure<readlane_pair_fma>(out, ticks, fma_ns);
  MeasureSingleWave<fma_f64>(out, ticks);
  MeasureSingleWave<fmac_f64_dpp>(out, ticks);
  MeasureSingleWave<add_u32>(out, ticks);
  MeasureSingleWave<readlane_b32>(out, ticks);
  MeasureSingleWave<readlane_pair_fma>(out, ticks);
  MeasureSingleWave<mov_b32_dpp>(out, ticks);
  MeasureLatency<fma_f64>(out, ticks);
  MeasureLatency<mul_f64>(out, ticks);
  MeasureLatency<add_f64>(out, ticks);
  MeasureLatency<fmac_f64>(out, ticks);
  MeasureLatency<fmac_f64_dpp>(out, ticks);
  MeasureLatency<dep_fmac_dpp_src>(out, ticks);
  MeasureLatency<dep_mov_dpp>(out, ticks);
  MeasureLatency<rsq_f64>(out, ticks);
  MeasureLatency<rcp_f64>(out, ticks);
  MeasureLatency<dep_readlane_fma>(out, ticks);
  MeasureLatency<add_u32>(out, ticks);
  return 0;
}

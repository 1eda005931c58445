// Where does global_load_lds_dwordx4 (gfx950) put lane L's 16 bytes?  (expected: M0 + 16 L)
// hipcc -O3 --offload-arch=gfx950 tools/lds_direct_probe.hip -o /tmp/ldsp && /tmp/ldsp
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const double* src, double* out) {
  __shared__ double big[16384];      // 128 KB: the second half starts above the 64 KB a 16-bit M0 field could address
  double* buf = big + (blockIdx.x ? 12288 : 0);
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += 64) big[i] = -1.0;
  __syncthreads();
  const double* g = src + lane * 2;      // lane L reads doubles 2L, 2L+1
  // (half-wave form used by the Cholesky chain: lanes 0..31 -> a row at buf + 8, lanes 32..63 -> a row at buf + 8 + 66, whose M0 is set 512 bytes early)
  if (lane < 32) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(buf + 8), 16, 0, 16);
  else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(buf + 8 + 66 - 64), 16, 0, 16);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[blockIdx.x * 1024 + i] = buf[i];
  if (blockIdx.x) for (int i = threadIdx.x; i < 512; i += 64) out[1024 + 512 + i] = big[(12288 * 8 % 65536) / 8 + i];      // where a wrapped address would land
}
int main() {
  double h[128], o[2048];
  for (int i = 0; i < 128; ++i) h[i] = i;
  double *d, *dout; (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&dout, sizeof(o));
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(2), dim3(64), 0, 0, d, dout);
  (void)hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  for (int b = 0; b < 3; ++b) { printf(b == 0 ? "target at LDS offset 8 x 8 bytes:\n" : (b == 1 ? "target at 96 KB + 64 bytes:\n" : "LDS at (96 KB mod 64 KB):\n")); for (int i = 0; i < 32; ++i) printf("%g%c", o[(b == 2 ? 1536 : b * 1024) + i], (i % 16 == 15) ? '\n' : ' '); }
  return 0;
}

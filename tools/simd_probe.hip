// Which SIMD does wavefront w of a 1024-thread workgroup land on? (HW_ID.SIMD_ID per wavefront)  hipcc --offload-arch=gfx950 tools/simd_probe.hip -o /tmp/simd && /tmp/simd
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k(unsigned* out) {
  __shared__ double pad[16000];
  pad[threadIdx.x] = 0;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main() {
  unsigned* d; (void)hipMalloc(&d, 4 * 16 * 4);
  hipLaunchKernelGGL(k, dim3(4), dim3(1024), 0, 0, d);
  unsigned h[64]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 4; ++b) { printf("wg %d:", b); for (int w = 0; w < 16; ++w) printf(" w%d:simd%u/wave%u/cu%u", w, (h[b*16+w] >> 4) & 3, h[b*16+w] & 15, (h[b*16+w] >> 8) & 15); printf("\n"); }
}

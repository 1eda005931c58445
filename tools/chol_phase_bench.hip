// Phase timing of the latency-bound panel kernels of K3b (k_potrf64 / k_step workgroup 0 / k_trsm64):
// stamps wall_clock64() (100 MHz) at the phase boundaries of ONE launch and times back-to-back launches.
// Build + run (GPU box):  hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/chol_phase_bench.hip \
//     privacy_preserving_sfm_amd/csrc/capi_misc.hip -o /tmp/chol_phase && /tmp/chol_phase
#define PP_CHOL_TRACE 1
#include "../privacy_preserving_sfm_amd/csrc/cholesky.hip"

#include <cstdio>
#include <random>

int main() {
  const int T = 8, N = T * 64;
  std::vector<double> h((size_t)N * N, 0.0);
  std::mt19937_64 rng(1);
  std::uniform_real_distribution<double> u(-1, 1);
  std::vector<double> B((size_t)N * N);
  for (auto& v : B) v = u(rng);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = 0;
      for (int k = 0; k < N; ++k) s += B[(size_t)i * N + k] * B[(size_t)j * N + k];
      h[(size_t)i * N + j] = s + (i == j ? N : 0);
    }
  double *S, *S0, *ws; int32_t* flag;
  hipMalloc(&S, sizeof(double) * N * N); hipMalloc(&S0, sizeof(double) * N * N); hipMalloc(&ws, sizeof(double) * N * 80); hipMalloc(&flag, 16);
  hipMemcpy(S0, h.data(), sizeof(double) * N * N, hipMemcpyHostToDevice);
  hipMemset(flag, 0, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto reset = [&]() { hipMemcpy(S, S0, sizeof(double) * N * N, hipMemcpyDeviceToDevice); hipDeviceSynchronize(); };
  long long tr[32];
  for (int rep = 0; rep < 3; ++rep) {
    reset();
    hipLaunchKernelGGL(ppsfm::k_potrf64, dim3(1), dim3(1024), 0, 0, S, N, 0, ws + (size_t)N * 64, flag, 0);
    hipLaunchKernelGGL(ppsfm::k_trsm64, dim3(4 * (T - 1)), dim3(256), 0, 0, S, N, 0, ws + (size_t)N * 64, 0);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(tr, HIP_SYMBOL(ppsfm::g_chol_trace), sizeof(tr));
    printf("potrf64(no update) phases [10ns ticks]:");
    for (int i = 1; i <= 11; ++i) printf(" %lld", tr[i] - tr[i - 1]);
    printf("  total %lld\n", tr[11] - tr[0]);
    printf("trsm64(no update) phases:");
    for (int i = 17; i <= 19; ++i) printf(" %lld", tr[i] - tr[i - 1]);
    printf("\n");
    hipLaunchKernelGGL(ppsfm::k_step, dim3(1), dim3(1024), 0, 0, S, N, 0, ws + (size_t)N * 64, flag);
    hipLaunchKernelGGL(ppsfm::k_trsm64, dim3(4 * (T - 2)), dim3(256), 0, 0, S, N, 1, ws + (size_t)N * 64, 1);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(tr, HIP_SYMBOL(ppsfm::g_chol_trace), sizeof(tr));
    printf("k_step WG0 (update+potrf) phases:");
    for (int i = 1; i <= 11; ++i) printf(" %lld", tr[i] - tr[i - 1]);
    printf("  total %lld\n", tr[11] - tr[0]);
    printf("trsm64(update) phases:");
    for (int i = 17; i <= 19; ++i) printf(" %lld", tr[i] - tr[i - 1]);
    printf("\n");
  }
  // back-to-back launch cost
  const int R = 200;
  reset();
  hipEventRecord(e0, 0);
  for (int r = 0; r < R; ++r) hipLaunchKernelGGL(ppsfm::k_potrf64, dim3(1), dim3(1024), 0, 0, S, N, 0, ws + (size_t)N * 64, flag, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("k_potrf64 back-to-back: %.2f us per launch\n", ms * 1e3 / R);
  hipEventRecord(e0, 0);
  for (int r = 0; r < R; ++r) hipLaunchKernelGGL(ppsfm::k_trsm64, dim3(4 * (T - 1)), dim3(256), 0, 0, S, N, 0, ws + (size_t)N * 64, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf("k_trsm64 back-to-back: %.2f us per launch\n", ms * 1e3 / R);
  return 0;
}

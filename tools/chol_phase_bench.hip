// Phase timing of the chain workgroup of K3b's k_column_step (and k_potrf64):
// stamps wall_clock64() (100 MHz) at the phase boundaries of ONE launch and times back-to-back launches.
// Build + run (GPU box):  hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/chol_phase_bench.hip \
//     privacy_preserving_sfm_amd/csrc/capi_misc.hip -o /tmp/chol_phase && /tmp/chol_phase
#define PP_CHOL_TRACE 1
#include "../privacy_preserving_sfm_amd/csrc/cholesky.hip"

#include <algorithm>
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
  double *S, *S0, *ws; int32_t* flag; int32_t* ctrp = nullptr;
  hipMalloc(&S, sizeof(double) * N * N); hipMalloc(&S0, sizeof(double) * N * N); hipMalloc(&ws, sizeof(double) * ppsfm::CholeskyWorkspaceDoubles(N)); hipMalloc(&flag, 16);
  hipMemcpy(S0, h.data(), sizeof(double) * N * N, hipMemcpyHostToDevice);
  hipMemset(flag, 0, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto reset = [&]() { hipMemcpy(S, S0, sizeof(double) * N * N, hipMemcpyDeviceToDevice); hipDeviceSynchronize(); };
  long long tr[32];
  const int order[] = {0, 1, 2, 12, 13, 3, 4, 5, 6, 7, 8, 9, 10, 11};
  const char* names[] = {"load", "-", "X M^T", "D col0 -= XX^T", "panel0 (+side jobs)", "trail0", "panel1", "trail1", "panel2", "trail2 (+M10)", "panel3", "post (inv3, M rows 2-3)", "store"};
  for (int rep = 0; rep < 3; ++rep) {
    reset();
    hipLaunchKernelGGL(ppsfm::k_potrf64, dim3(1), dim3(1024), 0, 0, S, N, ws, ws + (size_t)N * 64, flag, (double*)nullptr, S, ctrp, 0, (double*)nullptr, 0ll, ppsfm::OneChain(N / 64));
    hipLaunchKernelGGL(ppsfm::k_column_step, dim3(1), dim3(1024), 0, 0, S, N, 0, T, ws, ws + (size_t)N * 64, flag, 1 << 30, 1 << 30, (const int32_t*)nullptr, 0, 0, 0);   // chain workgroup only
    hipDeviceSynchronize();
    // a k >= 1 chain step (with the panel k-1 updates); run the bulk of step 0 first so that column 0 is solved
    reset();
    hipLaunchKernelGGL(ppsfm::k_potrf64, dim3(1), dim3(1024), 0, 0, S, N, ws, ws + (size_t)N * 64, flag, (double*)nullptr, S, ctrp, 0, (double*)nullptr, 0ll, ppsfm::OneChain(N / 64));
    hipLaunchKernelGGL(ppsfm::k_column_step, dim3(3 + T - 3), dim3(1024), 0, 0, S, N, 0, T, ws, ws + (size_t)N * 64, flag, 1 << 30, 1 << 30, (const int32_t*)nullptr, 0, 0, 0);
    hipLaunchKernelGGL(ppsfm::k_column_step, dim3(1), dim3(1024), 0, 0, S, N, 1, T, ws, ws + (size_t)N * 64, flag, 1 << 30, 1 << 30, (const int32_t*)nullptr, 0, 0, 0);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(tr, HIP_SYMBOL(ppsfm::g_chol_trace), sizeof(tr));
    printf("chain workgroup, k=1 [10 ns ticks]:");
    for (int i = 1; i < 14; ++i) printf(" %s %lld |", names[i - 1], tr[order[i]] - tr[order[i - 1]]);
    printf("  total %lld\n", tr[11] - tr[0]);
  }
  // the same stamps inside a FULL-size factorisation (T = 47, all workgroups present): launches 0..kt, stamps of launch kt
  {
    const int T2 = 47, N2 = T2 * 64;
    std::vector<double> h2((size_t)N2 * N2, 0.0);
    for (int i = 0; i < N2; ++i)
      for (int j = 0; j <= i; ++j) h2[(size_t)i * N2 + j] = (i == j ? N2 : 0.0) + 0.25 * u(rng);
    double *S2, *ws2;
    hipMalloc(&S2, sizeof(double) * N2 * N2); hipMalloc(&ws2, sizeof(double) * ppsfm::CholeskyWorkspaceDoubles(N2));
    for (int kt : {2, 20, 40}) {
      hipMemcpy(S2, h2.data(), sizeof(double) * N2 * N2, hipMemcpyHostToDevice);
      double* xs2 = ws2 + (size_t)N2 * 64;
      hipLaunchKernelGGL(ppsfm::k_potrf64, dim3(1), dim3(1024), 0, 0, S2, N2, ws2, xs2, flag, (double*)nullptr, S2, ctrp, 0, (double*)nullptr, 0ll, ppsfm::OneChain(N2 / 64));
      for (int k = 0; k <= kt; ++k) {
        const int n_prep = (k + 2 < T2) ? 2 : 0, nT = std::max(T2 - k - 3, 0), nb = T2 - k - 1;
        const int ns = (nb + 1) / 2, nsup = (k >= 1) ? ns * (ns + 1) / 2 - 1 : 0;
        const int nW = std::min(nsup, 4 * ppsfm::kNumCUs);
        hipLaunchKernelGGL(ppsfm::k_column_step, dim3(1 + n_prep + nT + nW), dim3(1024), 0, 0, S2, N2, k, T2, ws2, xs2, flag, 1 << 30, 1 << 30, (const int32_t*)nullptr, 0, 0, 0);
      }
      hipDeviceSynchronize();
      hipMemcpyFromSymbol(tr, HIP_SYMBOL(ppsfm::g_chol_trace), sizeof(tr));
      printf("full size T=47, chain of launch k=%d:", kt);
      for (int i = 1; i < 14; ++i) printf(" %s %lld |", names[i - 1], tr[order[i]] - tr[order[i - 1]]);
      printf("  total %lld\n", tr[11] - tr[0]);
      printf("   relative to the chain workgroup's entry: chain %lld..%lld | prepX %lld..%lld | prepD %lld..%lld | first trsm tile %lld..%lld | last syrk group %lld..%lld\n",
             tr[20] - tr[20], tr[21] - tr[20], tr[16] - tr[20], tr[17] - tr[20], tr[24] - tr[20], tr[25] - tr[20], tr[22] - tr[20], tr[23] - tr[20], tr[18] - tr[20],
             tr[19] - tr[20]);
    }
  }
  // the launch boundary as the chain sees it, in a complete factorisation: entry of launch k - exit of launch k-1, and how long
  // after the chain's exit the last workgroup of the launch leaves
  {
    const int T2 = 47, N2 = T2 * 64;
    std::vector<double> h2((size_t)N2 * N2, 0.0);
    for (int i = 0; i < N2; ++i)
      for (int j = 0; j <= i; ++j) h2[(size_t)i * N2 + j] = (i == j ? N2 : 0.0) + 0.25 * u(rng);
    double *S2, *ws2;
    hipMalloc(&S2, sizeof(double) * N2 * N2); hipMalloc(&ws2, sizeof(double) * ppsfm::CholeskyWorkspaceDoubles(N2));
    hipMemcpy(S2, h2.data(), sizeof(double) * N2 * N2, hipMemcpyHostToDevice);
    static long long zero[3][64] = {};
    hipMemcpyToSymbol(HIP_SYMBOL(ppsfm::g_chol_launch), zero, sizeof(zero));
    double* xs2 = ws2 + (size_t)N2 * 64;
    hipLaunchKernelGGL(ppsfm::k_potrf64, dim3(1), dim3(1024), 0, 0, S2, N2, ws2, xs2, flag, (double*)nullptr, S2, ctrp, 0, (double*)nullptr, 0ll, ppsfm::OneChain(N2 / 64));
    for (int k = 0; k + 1 < T2; ++k) {
      const int n_prep = (k + 2 < T2) ? 2 : 0, nT = std::max(T2 - k - 3, 0), nb = T2 - k - 1;
      const int ns = (nb + 1) / 2, nsup = (k >= 1) ? ns * (ns + 1) / 2 - 1 : 0;
      const int nW = std::min(nsup, 4 * ppsfm::kNumCUs);
      hipLaunchKernelGGL(ppsfm::k_column_step, dim3(1 + n_prep + nT + nW), dim3(1024), 0, 0, S2, N2, k, T2, ws2, xs2, flag, 1 << 30, 1 << 30, (const int32_t*)nullptr, 0, 0, 0);
    }
    hipDeviceSynchronize();
    // the same sequence with roles switched off (timing only): which role sets the cadence of the launches
    for (int mask : {0, 1, 2, 4, 8, 14, 15, 7, 11, 13}) {
      hipMemcpyToSymbol(HIP_SYMBOL(ppsfm::g_chol_skip), &mask, sizeof(int));
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(S2, h2.data(), sizeof(double) * N2 * N2, hipMemcpyHostToDevice);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int k = 10; k + 1 < T2; ++k) {
          const int n_prep = (k + 2 < T2) ? 2 : 0, nT = std::max(T2 - k - 3, 0), nb = T2 - k - 1;
          const int ns = (nb + 1) / 2, nsup = (k >= 1) ? ns * (ns + 1) / 2 - 1 : 0;
          const int nW = std::min(nsup, 4 * ppsfm::kNumCUs);
          hipLaunchKernelGGL(ppsfm::k_column_step, dim3(1 + n_prep + nT + nW), dim3(1024), 0, 0, S2, N2, k, T2, ws2, xs2, flag, 1 << 30, 1 << 30, (const int32_t*)nullptr, 0, 0, 0);
        }
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
      }
      printf("roles off mask %2d (1 chain, 2 prep, 4 solves, 8 trailing): launches k = 10..45: %.2f us per launch\n", mask, best * 1e3 / (T2 - 11));
    }
    { const int mask = 0; hipMemcpyToSymbol(HIP_SYMBOL(ppsfm::g_chol_skip), &mask, sizeof(int)); }
    static long long lt[3][64];
    hipMemcpyFromSymbol(lt, HIP_SYMBOL(ppsfm::g_chol_launch), sizeof(lt));
    printf("per launch [10 ns ticks]: k: chain entry - previous chain exit | chain duration | last workgroup exit - chain exit\n");
    for (int k = 1; k + 1 < T2; ++k) printf("  k=%d: %lld | %lld | %lld\n", k, lt[0][k] - lt[1][k - 1], lt[1][k] - lt[0][k], lt[2][k] - lt[1][k]);
  }
  // back-to-back launch cost
  const int R = 200;
  reset();
  hipEventRecord(e0, 0);
  for (int r = 0; r < R; ++r) hipLaunchKernelGGL(ppsfm::k_potrf64, dim3(1), dim3(1024), 0, 0, S, N, ws, ws + (size_t)N * 64, flag, (double*)nullptr, S, ctrp, 0, (double*)nullptr, 0ll, ppsfm::OneChain(N / 64));
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("k_potrf64 back-to-back: %.2f us per launch\n", ms * 1e3 / R);
  hipEventRecord(e0, 0);
  for (int r = 0; r < R; ++r) hipLaunchKernelGGL(ppsfm::k_column_step, dim3(1), dim3(1024), 0, 0, S, N, 1, T, ws, ws + (size_t)N * 64, flag, 1 << 30, 1 << 30, (const int32_t*)nullptr, 0, 0, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf("k_column_step (chain only) back-to-back: %.2f us per launch\n", ms * 1e3 / R);
  {      // per-wavefront arrival at the barriers of PotrfPanels (last chain step / launch), relative to the earliest arrival at barrier 0
    static long long wa[12][16];
    hipMemcpyFromSymbol(wa, HIP_SYMBOL(ppsfm::g_wave_arrive), sizeof(wa));
    long long base = wa[0][0];
    for (int w = 0; w < 16; ++w) base = std::min(base, wa[0][w]);
    printf("wavefront arrival at the PotrfPanels barriers [us after the first arrival at barrier 0]; barriers: side/panel0, trail0, panel1, trail1, panel2, trail2, panel3, last products\n");
    for (int w = 0; w < 16; ++w) {
      printf("w%2d |", w);
      for (int b = 0; b < 8; ++b) printf(" %6.2f", (wa[b][w] - base) * 0.01);
      printf("\n");
    }
  }
  return 0;
}

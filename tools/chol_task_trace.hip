// Timeline of ONE task-mode factorisation (k_cholesky_tasks) at cfg-3 size (T = 47): wall_clock64 stamps (100 MHz) per block
// column of the chain workgroup (wait begin / work begin / end), the prep tasks and the first / last bulk task.
// Build + run (GPU box):  hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/chol_task_trace.hip \
//     privacy_preserving_sfm_amd/csrc/capi_misc.hip -o /tmp/chol_task_trace && /tmp/chol_task_trace
#define PP_CHOL_TRACE 1
#ifndef PP_CHOL_SRC      // (A/B on one box against another revision of the file: -DPP_CHOL_SRC='"_ab/cholesky_base.inc"', see tools/ab_chain.sh)
#define PP_CHOL_SRC "../privacy_preserving_sfm_amd/csrc/cholesky.hip"
#endif
#include PP_CHOL_SRC

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 47, N = T * 64;
  std::mt19937_64 rng(1);
  std::uniform_real_distribution<double> u(-1, 1);
  std::vector<double> h((size_t)N * N, 0.0);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j <= i; ++j) h[(size_t)i * N + j] = (i == j ? N : 0.0) + 0.25 * u(rng);
  double *S, *L, *ws, *x; int32_t* flag;
  hipMalloc(&S, sizeof(double) * N * N); hipMalloc(&L, sizeof(double) * N * N); hipMalloc(&ws, sizeof(double) * ppsfm::CholeskyWorkspaceDoubles(N));
  hipMalloc(&x, sizeof(double) * N); hipMalloc(&flag, 16);
  hipMemset(flag, 0, 16);
  ppsfm::CholeskyAux aux;
  aux.mode = 1; aux.use_graph = false;
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  static long long tr[24][128];
  if (getenv("PP_ARRIVE_STEP")) { const int st = atoi(getenv("PP_ARRIVE_STEP")); hipMemcpyToSymbol(HIP_SYMBOL(ppsfm::g_arrive_step), &st, sizeof(st)); }      // barrier arrivals of THAT step instead of the last one
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(S, h.data(), sizeof(double) * N * N, hipMemcpyHostToDevice);
    for (int a = 0; a < 24; ++a) for (int k = 0; k < 128; ++k) tr[a][k] = (a == 7 || a == 10) ? (1ll << 62) : 0;
    hipMemcpyToSymbol(HIP_SYMBOL(ppsfm::g_task_trace), tr, sizeof(tr));
    hipDeviceSynchronize();
    hipEventRecord(e0, s);
    ppsfm::CholeskySolveAugmented(S, N, N - 1, ws, L, x, flag, s, &aux);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int32_t f[4]; hipMemcpy(f, flag, 16, hipMemcpyDeviceToHost);
    printf("rep %d: %.1f us, flag %d\n", rep, ms * 1e3, f[0]);
  }
  if (argc > 2 && argv[2][0] == 'c') {      // check: the factor L against a host Cholesky, worst relative error per 64x64 tile
    std::vector<double> ref(h), got((size_t)N * N);
    for (int j = 0; j < N; ++j) {
      double d = ref[(size_t)j * N + j];
      for (int k = 0; k < j; ++k) d -= ref[(size_t)j * N + k] * ref[(size_t)j * N + k];
      d = std::sqrt(d); ref[(size_t)j * N + j] = d;
      for (int i = j + 1; i < N; ++i) {
        double v = ref[(size_t)i * N + j];
        for (int k = 0; k < j; ++k) v -= ref[(size_t)i * N + k] * ref[(size_t)j * N + k];
        ref[(size_t)i * N + j] = v / d;
      }
    }
    hipMemcpy(got.data(), L, sizeof(double) * N * N, hipMemcpyDeviceToHost);
    printf("tile errors (rows = block row, '.' < 1e-9, digit = -log10 of the error otherwise; diagonal blocks are not stored in task mode)\n");
    for (int bi = 0; bi < T; ++bi) {
      for (int bj = 0; bj < bi; ++bj) {
        double e = 0;
        for (int r = 0; r < 64; ++r) for (int cc = 0; cc < 64; ++cc) {
          const double a = got[(size_t)(bi * 64 + r) * N + bj * 64 + cc], b = ref[(size_t)(bi * 64 + r) * N + bj * 64 + cc];
          const double dd = std::fabs(a - b);
          e = std::max(e, (dd == dd) ? dd : 1e300);
        }
        if (e < 1e-9) putchar('.'); else { int dg = (int)std::floor(-std::log10(e)); putchar(dg < 0 ? 'X' : (dg > 9 ? '9' : '0' + dg)); }
      }
      putchar('\n');
    }
  }
  if (getenv("PP_BS_TRACE")) {      // the back substitution's timeline: per single block / pair (listed by its lowest block)
    static long long bs[5][128];
    hipMemcpyFromSymbol(bs, HIP_SYMBOL(ppsfm::g_bs_trace), sizeof(bs));
    long long t00 = 1ll << 62;
    for (int j = 0; j < T; ++j) if (bs[0][j]) t00 = std::min(t00, bs[0][j]);
    printf("back substitution [us after the first entry]: block | entry | far terms done | newest input seen | published | since the previous publication\n");
    double prev = 0;
    for (int j = T - 1; j >= 0; --j) {
      if (!bs[3][j]) continue;
      const double pub = (bs[3][j] - t00) * 0.01;
      printf("%2d | %6.2f %6.2f (G^T u done %6.2f) %6.2f %6.2f | %5.2f\n", j, (bs[0][j] - t00) * 0.01, bs[1][j] ? (bs[1][j] - t00) * 0.01 : -1.0, bs[4][j] ? (bs[4][j] - t00) * 0.01 : -1.0, bs[2][j] ? (bs[2][j] - t00) * 0.01 : -1.0, pub, pub - prev);
      prev = pub;
    }
  }
  hipMemcpyFromSymbol(tr, HIP_SYMBOL(ppsfm::g_task_trace), sizeof(tr));
  const long long t0 = tr[1][0];
  for (int k = 0; k < 128; ++k) tr[0][k] = tr[1][k];      // (the chain no longer has a separate wait phase)
  auto us = [&](long long t) { return t ? (t - t0) * 0.01 : -1.0; };
  printf("[us from the chain's first stamp]\n k | chain: wait-begin work-begin end (work, wait) | prepX: A-begin B-wait B-begin end | prepD: A-begin B-wait B-begin end | solves: first-start last-end | updates: first-start last-end\n");
  for (int k = 0; k + 1 < T; ++k) {
    const bool has7 = tr[7][k] < (1ll << 61), has10 = tr[10][k] < (1ll << 61);
    printf("%2d | %7.1f %7.1f %7.1f (%5.1f, %5.1f) | %7.1f %7.1f %7.1f %7.1f | %7.1f %7.1f %7.1f %7.1f | %7.1f %7.1f | %7.1f %7.1f\n", k, us(tr[0][k]), us(tr[1][k]), us(tr[2][k]),
           us(tr[2][k]) - us(tr[1][k]), us(tr[1][k]) - us(tr[0][k]), us(tr[3][k]), us(tr[12][k]), us(tr[4][k]), us(tr[5][k]), us(tr[11][k]), us(tr[13][k]), us(tr[14][k]), us(tr[6][k]),
           has7 ? us(tr[7][k]) : -1.0, us(tr[9][k]), has10 ? us(tr[10][k]) : -1.0, us(tr[8][k]));
  }
  {
    static long long sw[3][128];
    hipMemcpyFromSymbol(sw, HIP_SYMBOL(ppsfm::g_spare_wait), sizeof(sw));
    printf("chain, per step: wait of the spare wavefronts after the last panel [us] (state of the X / D fetch when they got there: 1 = in flight, 2 = in LDS) | step length [us]\n");
    double total = 0;
    for (int k = 0; k + 2 < T; ++k) { printf("%2d | %5.2f (X %lld, D %lld) | %6.2f\n", k, sw[0][k] * 0.01, sw[1][k] / 4, sw[1][k] % 4, (sw[2][k + 1] - sw[2][k]) * 0.01); total += sw[0][k] * 0.01; }
    printf("total wait %.1f us\n", total);
  }
  printf(" k | front update of step k (the super-tiles PrepX(k+1) / PrepD(k+1) wait for): (I, J=(k+1)/2) entry, start, end | (I, J=(k+3)/2) entry, start, end | PrepX(k) entry | PrepD(k) entry\n");
  static unsigned long long wm[128];
  hipMemcpyFromSymbol(wm, HIP_SYMBOL(ppsfm::g_wait_missing), sizeof(wm));
  for (int k = 1; k + 1 < T; ++k) printf("%2d | missing in the last round 0x%02llx after %llu rounds | %7.1f %7.1f %7.1f | %7.1f %7.1f %7.1f | %7.1f | %7.1f\n", k, wm[k] & 255, wm[k] >> 8, us(tr[16][k]), us(tr[17][k]), us(tr[18][k]), us(tr[21][k]), us(tr[22][k]), us(tr[23][k]), us(tr[19][k]), us(tr[20][k]));
  {
    long long ct[32];
    hipMemcpyFromSymbol(ct, HIP_SYMBOL(ppsfm::g_chol_trace), sizeof(ct));
    printf("chain, LAST step, inside the panels [us]: panel0 (+side) %.2f | trail0 %.2f | panel1 (+side1) %.2f | trail1 %.2f | panel2 %.2f | trail2 %.2f | panel3 %.2f\n",
           0.0, (ct[4] - ct[3]) * 0.01, (ct[5] - ct[4]) * 0.01, (ct[6] - ct[5]) * 0.01, (ct[7] - ct[6]) * 0.01, (ct[8] - ct[7]) * 0.01, (ct[9] - ct[8]) * 0.01);
  }
  {
    static long long c2[32][128];
    static long long ph2[8][128];
    hipMemcpyFromSymbol(c2, HIP_SYMBOL(ppsfm::g_chol_trace2), sizeof(c2));
    hipMemcpyFromSymbol(ph2, HIP_SYMBOL(ppsfm::g_chain_phase), sizeof(ph2));
    static long long clk[128];
    hipMemcpyFromSymbol(clk, HIP_SYMBOL(ppsfm::g_chain_clk), sizeof(clk));
    printf("shader clock of the chain's CU [MHz] over steps 1-5, 20-25, 40-44: %.0f %.0f %.0f\n", (clk[5] - clk[1]) / ((ph2[0][5] - ph2[0][1]) * 0.01),
           (clk[25] - clk[20]) / ((ph2[0][25] - ph2[0][20]) * 0.01), (clk[44] - clk[40]) / ((ph2[0][44] - ph2[0][40]) * 0.01));
    printf("chain, inside the panels per step [us]: panel0 (+side) | trail0 | panel1 (+side1) | trail1 | panel2 | trail2 | panel3 | post\n");
    for (int k = 0; k + 2 < T; k += 4)
      printf("%2d | %5.2f %5.2f %5.2f %5.2f %5.2f %5.2f %5.2f %5.2f\n", k, (c2[3][k] - ph2[4][k]) * 0.01, (c2[4][k] - c2[3][k]) * 0.01, (c2[5][k] - c2[4][k]) * 0.01, (c2[6][k] - c2[5][k]) * 0.01,
             (c2[7][k] - c2[6][k]) * 0.01, (c2[8][k] - c2[7][k]) * 0.01, (c2[9][k] - c2[8][k]) * 0.01, (ph2[5][k] - c2[9][k]) * 0.01);
  }
  static long long ph[8][128];
  hipMemcpyFromSymbol(ph, HIP_SYMBOL(ppsfm::g_chain_phase), sizeof(ph));
  printf("chain workgroup, thread 0, per step [us]: barrier-in | solve | solve-barrier (stores acknowledged, publish) | D col 0 | panels | store issue | -> next step's first stamp\n");
  for (int k = 0; k + 2 < T; ++k)
    printf("%2d | %5.2f %5.2f %5.2f %5.2f %6.2f %5.2f %5.2f\n", k, (ph[1][k] - ph[0][k]) * 0.01, (ph[2][k] - ph[1][k]) * 0.01, (ph[3][k] - ph[2][k]) * 0.01, (ph[4][k] - ph[3][k]) * 0.01,
           (ph[5][k] - ph[4][k]) * 0.01, (ph[6][k] - ph[5][k]) * 0.01, (ph[0][k + 1] - ph[6][k]) * 0.01);
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
  if (argc > 3) {      // the chain workgroup ALONE: mailboxes still hold the previous run's X / D / M tiles, so it never waits
    using namespace ppsfm;
    const size_t tile = 64 * 64;
    Mailboxes mb;
    mb.Minv = ws; mb.xs = ws + (size_t)T * tile; mb.ds = mb.xs + (size_t)(T + 1) * tile; mb.xsol = mb.ds + (size_t)(T + 1) * tile;
    int32_t* ctr = reinterpret_cast<int32_t*>(mb.xsol + (size_t)(T + 1) * tile);
    const int nburn_arg = argc > 4 ? atoi(argv[4]) : 0;
    const int exps[] = {1, 2, 3, 4, 7, 8, 16, 0, 0, 0, 0, 0, 0};
    const int burns[] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 7, 63, 255, nburn_arg};
    for (int rep = 0; rep < 13; ++rep) {
      const int e = exps[rep];
      const int nburn = burns[rep];
      { const int z = 0; hipMemcpyToSymbol(HIP_SYMBOL(ppsfm::g_burn_stop), &z, sizeof(z)); }
      hipMemcpyToSymbol(HIP_SYMBOL(ppsfm::g_chol_exp), &e, sizeof(e));
      hipMemcpy(S, h.data(), sizeof(double) * N * N, hipMemcpyHostToDevice);
      hipDeviceSynchronize();
      hipLaunchKernelGGL(k_potrf64, dim3(1), dim3(kPanelThreads), 0, s, S, N, ws, mb.xs, flag, x, L, ctr, (int)kNumCounters, ws, (long long)((size_t)(4 * T + 3) * tile), OneChain(T));
      hipEventRecord(e0, s);
      hipLaunchKernelGGL(k_cholesky_tasks, dim3(1 + nburn), dim3(kPanelThreads), 0, s, S, L, N, T, mb, flag, ctr, (const ChainTask*)nullptr, (const uint8_t*)nullptr, OneChain(T), (double*)nullptr);
      hipEventRecord(e1, s); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("chain alone, switches %2d (1: no X->mailbox, 2: no X->L, 4: no M->mailbox, 8: fetch only after the panels, 16: fetched tiles compared with their mailboxes): %.1f us = %.2f us per step\n", e, ms * 1e3, ms * 1e3 / (T - 1));
      static long long wa[12][16];
      hipMemcpyFromSymbol(wa, HIP_SYMBOL(ppsfm::g_wave_arrive), sizeof(wa));
      long long base = wa[0][0];
      for (int w = 0; w < 16; ++w) base = std::min(base, wa[0][w]);
      if (e == 0 && rep == 8) {
        printf("   per wavefront: arrival at the step's first four barriers (step start, after the solve, after the X store, after D column 0) [us before the first arrival at the panel-0 barrier]\n");
        for (int w = 0; w < 16; ++w) printf("   w%2d | %6.2f %6.2f %6.2f %6.2f\n", w, (wa[8][w] - base) * 0.01, (wa[9][w] - base) * 0.01, (wa[10][w] - base) * 0.01, (wa[11][w] - base) * 0.01);
      }
      if (nburn > 0) {
        static unsigned hw[512];
        hipMemcpyFromSymbol(hw, HIP_SYMBOL(ppsfm::g_burn_hwid), sizeof(hw));
        auto cu = [](unsigned v) { return (v >> 8) & 15; }; auto sh = [](unsigned v) { return (v >> 12) & 1; }; auto se = [](unsigned v) { return (v >> 13) & 7; }; auto xc = [](unsigned v) { return v >> 16; };
        int same_pair = 0, same_sa = 0, same_xcc = 0;
        for (int i = 1; i <= nburn && i < 512; ++i) {
          if (xc(hw[i]) == xc(hw[0])) { ++same_xcc; if (se(hw[i]) == se(hw[0]) && sh(hw[i]) == sh(hw[0])) { ++same_sa; if ((cu(hw[i]) ^ 1) == cu(hw[0])) ++same_pair; } }
        }
        printf("   %d busy workgroups beside the chain (chain on xcc %u se %u sh %u cu %u): %d on its XCC, %d in its shader array, %d on the CU whose id differs in bit 0\n", nburn, xc(hw[0]), se(hw[0]), sh(hw[0]), cu(hw[0]), same_xcc, same_sa, same_pair);
      }
      if (e == 16) { int mm[16]; hipMemcpyFromSymbol(mm, HIP_SYMBOL(ppsfm::g_dbg_mismatch), sizeof(mm)); int tot = 0; for (int i = 0; i < 16; ++i) tot += mm[i]; printf("   fetched-tile mismatches: %d\n", tot); }
      printf("   w0 at the barriers:");
      for (int b = 0; b < 8; ++b) printf(" %6.2f", (wa[b][0] - base) * 0.01);
      printf("   latest:");
      for (int b = 0; b < 8; ++b) { long long m = 0; for (int w = 0; w < 16; ++w) m = std::max(m, wa[b][w]); printf(" %6.2f", (m - base) * 0.01); }
      printf("\n");
    }
  }
  return 0;
}

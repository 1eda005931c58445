// Can two kernels of one hipGraph (parallel branches) or of two streams run CONCURRENTLY on MI355X, and what does a flag
// hand-off between them cost?  (Design probe for taking the Cholesky chain workgroup out of the per-column launches.)
//   waiter: one workgroup, spins (bounded) on flag[i] for i = 0..R-1, answering each in ack[i]  -> stamps the round trip
//   poker : one workgroup, for i = 0..R-1: sets flag[i], spins (bounded) on ack[i]
// If the two kernels are serialised, the first one runs into its spin bound and reports it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int kRounds = 64;
constexpr int kSpinBound = 1 << 22;     // ~ a few hundred ms at most

__global__ void k_waiter(int* flag, int* ack, long long* stamps, int* fail) {
  if (threadIdx.x != 0) return;
  for (int i = 0; i < kRounds; ++i) {
    int spins = 0;
    while (__hip_atomic_load(flag + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && spins < kSpinBound) { __builtin_amdgcn_s_sleep(1); ++spins; }
    if (spins >= kSpinBound) { atomicOr(fail, 1); return; }
    __hip_atomic_store(ack + i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ void k_poker(int* flag, int* ack, long long* stamps, int* fail) {
  if (threadIdx.x != 0) return;
  for (int i = 0; i < kRounds; ++i) {
    const long long t0 = wall_clock64();
    __hip_atomic_store(flag + i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(ack + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && spins < kSpinBound) { __builtin_amdgcn_s_sleep(1); ++spins; }
    if (spins >= kSpinBound) { atomicOr(fail, 2); return; }
    stamps[i] = wall_clock64() - t0;
  }
}
// filler: occupies `blocks` workgroups for ~`ticks` x 10 ns (the bulk launches that would run beside the chain kernel)
__global__ void k_filler(long long ticks, double* sink) {
  const long long t0 = wall_clock64();
  double x = threadIdx.x;
  while (wall_clock64() - t0 < ticks) x = fma(x, 1.0000001, 1e-9);
  if (x == 12345.0) sink[0] = x;
}

static void Report(const char* what, int* d_fail, long long* d_stamps) {
  int fail; long long st[kRounds];
  CK(hipMemcpy(&fail, d_fail, sizeof(int), hipMemcpyDeviceToHost));
  CK(hipMemcpy(st, d_stamps, sizeof(st), hipMemcpyDeviceToHost));
  if (fail) { printf("%-44s SERIALISED (spin bound hit, code %d)\n", what, fail); return; }
  long long lo = 1 << 30, sum = 0;
  for (int i = 8; i < kRounds; ++i) { lo = st[i] < lo ? st[i] : lo; sum += st[i]; }
  printf("%-44s concurrent; flag round trip (two hops) min %.2f us, mean %.2f us\n", what, lo * 0.01, sum * 0.01 / (kRounds - 8));
}

int main() {
  int *flag, *ack, *fail; long long* stamps; double* sink;
  CK(hipMalloc(&flag, kRounds * 4)); CK(hipMalloc(&ack, kRounds * 4)); CK(hipMalloc(&fail, 4)); CK(hipMalloc(&stamps, kRounds * 8)); CK(hipMalloc(&sink, 8));
  auto reset = [&]() { CK(hipMemset(flag, 0, kRounds * 4)); CK(hipMemset(ack, 0, kRounds * 4)); CK(hipMemset(fail, 0, 4)); CK(hipMemset(stamps, 0, kRounds * 8)); CK(hipDeviceSynchronize()); };
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  // 1. two streams, waiter first
  reset();
  hipLaunchKernelGGL(k_waiter, dim3(1), dim3(64), 0, sa, flag, ack, stamps, fail);
  hipLaunchKernelGGL(k_poker, dim3(1), dim3(64), 0, sb, flag, ack, stamps, fail);
  CK(hipDeviceSynchronize());
  Report("two streams (waiter launched first)", fail, stamps);
  // 2. two streams, with 40 filler launches of 300 workgroups behind the poker on its stream (the bulk launches)
  reset();
  hipLaunchKernelGGL(k_waiter, dim3(1), dim3(64), 0, sa, flag, ack, stamps, fail);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_filler, dim3(300), dim3(1024), 0, sb, 1000LL, sink);
  hipLaunchKernelGGL(k_poker, dim3(1), dim3(64), 0, sb, flag, ack, stamps, fail);
  CK(hipDeviceSynchronize());
  Report("two streams, waiter resident under 5 full-chip launches", fail, stamps);
  // 3. one hipGraph with two parallel branches (captured from the two streams)
  for (int rep = 0; rep < 2; ++rep) {
    reset();
    hipGraph_t graph; hipGraphExec_t exec; hipEvent_t fork, join;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
    CK(hipEventRecord(fork, sa)); CK(hipStreamWaitEvent(sb, fork, 0));
    hipLaunchKernelGGL(k_waiter, dim3(1), dim3(64), 0, sa, flag, ack, stamps, fail);
    if (rep == 1) for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_filler, dim3(300), dim3(1024), 0, sb, 1000LL, sink);
    hipLaunchKernelGGL(k_poker, dim3(1), dim3(64), 0, sb, flag, ack, stamps, fail);
    CK(hipEventRecord(join, sb)); CK(hipStreamWaitEvent(sa, join, 0));
    CK(hipStreamEndCapture(sa, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(exec, sa));
    CK(hipDeviceSynchronize());
    Report(rep == 0 ? "hipGraph, two parallel branches" : "hipGraph, branches, 5 full-chip launches first", fail, stamps);
    // replay
    reset();
    CK(hipGraphLaunch(exec, sa));
    CK(hipDeviceSynchronize());
    Report(rep == 0 ? "  replay" : "  replay", fail, stamps);
    CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
  }
  return 0;
}

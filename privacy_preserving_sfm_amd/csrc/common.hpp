// Shared host/device helpers for the gfx950 library (libppsfm_hip.so).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <exception>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ppsfm_hip.h"

namespace ppsfm {

// thread-local last error string behind pp_last_error()
void SetLastError(const char* fmt, ...);
const char* LastError();

#define PP_HIP_TRY(expr)                                                                     \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      ::ppsfm::SetLastError("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                            __LINE__);                                                       \
      return PP_ERR_HIP;                                                                     \
    }                                                                                        \
  } while (0)

#define PP_REQUIRE(cond, ...)           \
  do {                                  \
    if (!(cond)) {                      \
      ::ppsfm::SetLastError(__VA_ARGS__); \
      return PP_ERR_INVALID;            \
    }                                   \
  } while (0)

// "never abort/throw across the boundary" (include/ppsfm_hip.h; the reference itself aborts through CHECK,
// src/optim/bundle_adjustment.cc:261-262): every extern "C" entry point that can allocate is a function-try-block
//     int pp_xxx(args) try { ... } PP_API_CATCH("pp_xxx")
// (the handler is part of the function, so it also covers a body that returns early through PP_REQUIRE / PP_HIP_TRY).
int ApiExceptionToCode(const char* where);      // called inside a catch (...) block: rethrows, classifies, sets pp_last_error(); never throws
#define PP_API_CATCH(where) catch (...) { return ::ppsfm::ApiExceptionToCode(where); }

// fn() when the scope is left by an EXCEPTION (a handle under construction, the scratch buffers of an entry point): the error
// returns of the TRY macros clean up themselves, a throw from a std::vector in between would otherwise leak device memory.
template <typename F>
struct OnUnwind {
  F fn;
  int n = std::uncaught_exceptions();
  ~OnUnwind() { if (std::uncaught_exceptions() > n) fn(); }
};
template <typename F> OnUnwind(F) -> OnUnwind<F>;

// body(t) for t = 0 .. nthreads - 1 on that many host threads (t = 0 on the caller's).  An exception inside a worker would end the
// process (std::terminate); here every worker hands its exception to the caller, all threads are joined whatever happens - also
// when std::thread's own constructor throws std::system_error half way - and the first exception is rethrown on the calling thread,
// where the entry point's PP_API_CATCH turns it into an error code.
template <typename Body>
inline void ParallelFor(int nthreads, Body&& body) {
  if (nthreads <= 1) { body(0); return; }
  std::vector<std::exception_ptr> err((size_t)nthreads);
  std::vector<std::thread> th;
  struct Joiner { std::vector<std::thread>& t; ~Joiner() { for (auto& x : t) if (x.joinable()) x.join(); } } joiner{th};
  th.reserve((size_t)nthreads - 1);
  for (int t = 1; t < nthreads; ++t)
    th.emplace_back([&err, &body, t]() { try { body(t); } catch (...) { err[(size_t)t] = std::current_exception(); } });
  try { body(0); } catch (...) { err[0] = std::current_exception(); }
  for (auto& x : th) x.join();
  for (auto& e : err) if (e) std::rethrow_exception(e);
}

constexpr int kWave = 64;          // gfx950 wavefront
constexpr int kCamStride = 12;     // doubles per intrinsics block (max kNumParams of the 11 models)

template <typename T>
inline int DeviceAlloc(T** p, size_t count) {
  *p = nullptr;
  if (count == 0) return PP_OK;
  PP_HIP_TRY(hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
  return PP_OK;
}
template <typename T>
inline int Upload(T* dst, const T* src, size_t count, hipStream_t s) {
  if (count == 0) return PP_OK;
  PP_HIP_TRY(hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyHostToDevice, s));
  return PP_OK;
}
template <typename T>
inline int Download(T* dst, const T* src, size_t count, hipStream_t s) {
  if (count == 0) return PP_OK;
  PP_HIP_TRY(hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyDeviceToHost, s));
  return PP_OK;
}

inline int CeilDiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// deterministic wave-level sum (butterfly over 64 lanes; every lane ends with the total)
__device__ __forceinline__ double WaveSum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// The same total without the LDS crossbar (ds_bpermute: two per step and double, and four wavefronts of a workgroup queue up on
// it): quad / half-row / row exchanges as DPP moves (VALU rate), the four row totals by v_readlane.  A different, equally fixed
// order - for the sums that are not pinned to WaveSum's butterfly by a bitwise test.
template <int kCtrl>
__device__ __forceinline__ double DppMove(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, kCtrl, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, kCtrl, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double WaveSumDpp(double v) {
  v += DppMove<0xB1>(v);       // quad_perm [1,0,3,2]
  v += DppMove<0x4E>(v);       // quad_perm [2,3,0,1]
  v += DppMove<0x141>(v);      // row_half_mirror
  v += DppMove<0x140>(v);      // row_mirror: every lane holds the sum of its 16-lane row
  int lo = __double2loint(v), hi = __double2hiint(v);
  const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
  const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
  const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
  const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
  return (r0 + r1) + (r2 + r3);
}

// sum / maximum over each 16-lane row (every lane of a row ends with its row's value): the DPP part of the above
__device__ __forceinline__ double RowSumDpp(double v) {
  v += DppMove<0xB1>(v); v += DppMove<0x4E>(v); v += DppMove<0x141>(v); v += DppMove<0x140>(v);
  return v;
}
__device__ __forceinline__ double RowMaxDpp(double v) {
  v = fmax(v, DppMove<0xB1>(v)); v = fmax(v, DppMove<0x4E>(v)); v = fmax(v, DppMove<0x141>(v)); v = fmax(v, DppMove<0x140>(v));
  return v;
}
// the maximum of a non-negative (or any NaN-free) value over the wavefront, every lane ends with it: the same moves as WaveSumDpp
__device__ __forceinline__ double WaveMaxDpp(double v) {
  v = fmax(v, DppMove<0xB1>(v));
  v = fmax(v, DppMove<0x4E>(v));
  v = fmax(v, DppMove<0x141>(v));
  v = fmax(v, DppMove<0x140>(v));
  int lo = __double2loint(v), hi = __double2hiint(v);
  const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
  const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
  const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
  const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
  return fmax(fmax(r0, r1), fmax(r2, r3));
}

}  // namespace ppsfm

// Shared host/device helpers for the gfx950 library (libppsfm_hip.so).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

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

}  // namespace ppsfm

// Recycled HIP resources of the bundle-adjustment handles: device blocks, pinned host blocks, streams and events.
//
// The mapper builds a NEW BundleAdjuster for every image it registers (src/sfm/incremental_mapper.cc:813-858, 6 images; the global one at
// controllers/incremental_mapper.cc:497-504 as the model grows): pp_ba_create / pp_ba_destroy sit in its inner loop.  Measured on MI355X for a
// 6-image / 2004-observation problem: create 0.40 ms, destroy 0.75 ms against 2.2 ms for a 25-iteration solve - ~70 hipMalloc / hipFree (a hipFree
// synchronises the device), a pinned allocation, a stream and 13 events per handle.  Blocks are kept by (device, size class) and handed out
// again; a handle synchronises its stream before it returns anything, so a block is never reused under a kernel that still reads it.
// PPSFM_POOL_MAX_MB (default 1024) caps the cached device bytes per process - beyond it a returned block is freed at once; 0 disables the pool.
#pragma once
#include "common.hpp"

namespace ppsfm {

int PoolDeviceAlloc(void** p, size_t bytes);      // on the CURRENT device
void PoolDeviceFree(void* p);                     // (a pointer the pool does not know is hipFree'd)
int PoolPinnedAlloc(void** p, size_t bytes);
void PoolPinnedFree(void* p);
int PoolStreamAcquire(hipStream_t* s);            // non-blocking stream of the current device
void PoolStreamRelease(hipStream_t s);
int PoolEventAcquire(hipEvent_t* e, bool timing);
void PoolEventRelease(hipEvent_t e, bool timing);
void PoolTrim();                                  // frees everything cached (tests; pp_pool_trim)

template <typename T>
inline int HandleAlloc(T** p, size_t count) {
  *p = nullptr;
  if (count == 0) return PP_OK;
  return PoolDeviceAlloc(reinterpret_cast<void**>(p), count * sizeof(T));
}

}  // namespace ppsfm

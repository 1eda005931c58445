// see resource_pool.hpp
#include "resource_pool.hpp"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace ppsfm {
namespace {

struct Block { int device; size_t bytes; };
struct Pool {
  std::mutex mu;
  std::unordered_map<void*, Block> live;                                   // device blocks handed out
  std::unordered_map<unsigned long long, std::vector<void*>> free_dev;     // (device << 48 | class index) -> blocks
  std::unordered_map<void*, size_t> live_pinned;
  std::unordered_map<size_t, std::vector<void*>> free_pinned;
  std::unordered_map<int, std::vector<hipStream_t>> streams;
  std::unordered_map<int, std::vector<hipEvent_t>> events[2];
  std::unordered_map<void*, int> stream_device, event_device;
  std::unordered_map<int, size_t> cached_bytes;      // per device (the cap is per device: a busy GPU 0 does not evict GPU 1's blocks)
  size_t max_bytes = 0;
  bool enabled = true;
  bool poison = false;      // PPSFM_POOL_POISON=1 (debug): every block handed out - fresh or recycled - is filled with 0xFF bytes (NaN doubles, -1 ints), so a
                            // read-before-write sees garbage instead of whatever the previous handle left (recycled) or zeros (a fresh hipMalloc often is)
  Pool() {
    const char* e = std::getenv("PPSFM_POOL_MAX_MB");
    const long mb = e ? std::atol(e) : 1024;
    enabled = mb > 0;
    max_bytes = (size_t)(mb > 0 ? mb : 0) << 20;
    const char* pz = std::getenv("PPSFM_POOL_POISON");
    poison = pz && std::atoi(pz) != 0;
  }
};
Pool& P() { static Pool* p = new Pool(); return *p; }      // (never destroyed: handles may outlive static destruction order)

// size classes: powers of two up to 1 MiB, multiples of 1 MiB above
size_t ClassBytes(size_t bytes) {
  if (bytes <= 256) return 256;
  if (bytes > ((size_t)1 << 20)) return (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
  size_t c = 256;
  while (c < bytes) c <<= 1;
  return c;
}
unsigned long long Key(int device, size_t cls) { return ((unsigned long long)device << 48) ^ (unsigned long long)cls; }

}  // namespace

int PoolDeviceAlloc(void** p, size_t bytes) {
  *p = nullptr;
  Pool& pool = P();
  int device = 0;
  PP_HIP_TRY(hipGetDevice(&device));
  const size_t cls = ClassBytes(bytes);
  if (pool.enabled) {
    std::lock_guard<std::mutex> lock(pool.mu);
    auto it = pool.free_dev.find(Key(device, cls));
    if (it != pool.free_dev.end() && !it->second.empty()) {
      *p = it->second.back(); it->second.pop_back();
      pool.cached_bytes[device] -= cls;
      pool.live[*p] = Block{device, cls};
      if (pool.poison) { PP_HIP_TRY(hipMemset(*p, 0xFF, cls)); PP_HIP_TRY(hipStreamSynchronize(nullptr)); }      // (a device memset may return before it has run; the handle's stream does not wait for the null stream)
      return PP_OK;
    }
  }
  hipError_t e = hipMalloc(p, cls);
  if (e != hipSuccess && pool.enabled) {      // out of memory with blocks cached: give them back and try once more
    (void)hipGetLastError();
    PoolTrim();
    e = hipMalloc(p, cls);
  }
  if (e != hipSuccess) { SetLastError("hipMalloc(%zu bytes) failed: %s", cls, hipGetErrorString(e)); *p = nullptr; return PP_ERR_HIP; }
  if (pool.enabled) { std::lock_guard<std::mutex> lock(pool.mu); pool.live[*p] = Block{device, cls}; }
  if (pool.poison) { PP_HIP_TRY(hipMemset(*p, 0xFF, cls)); PP_HIP_TRY(hipStreamSynchronize(nullptr)); }
  return PP_OK;
}

void PoolDeviceFree(void* p) {
  if (!p) return;
  Pool& pool = P();
  {
    std::lock_guard<std::mutex> lock(pool.mu);
    auto it = pool.live.find(p);
    if (it != pool.live.end()) {
      const Block b = it->second;
      pool.live.erase(it);
      if (pool.enabled && pool.cached_bytes[b.device] + b.bytes <= pool.max_bytes) {
        pool.free_dev[Key(b.device, b.bytes)].push_back(p);
        pool.cached_bytes[b.device] += b.bytes;
        return;
      }
    }
  }
  (void)hipFree(p);
}

int PoolPinnedAlloc(void** p, size_t bytes) {
  *p = nullptr;
  Pool& pool = P();
  const size_t cls = ClassBytes(bytes);
  if (pool.enabled) {
    std::lock_guard<std::mutex> lock(pool.mu);
    auto it = pool.free_pinned.find(cls);
    if (it != pool.free_pinned.end() && !it->second.empty()) {
      *p = it->second.back(); it->second.pop_back();
      pool.live_pinned[*p] = cls;
      if (pool.poison) std::memset(*p, 0xFF, cls);
      return PP_OK;
    }
  }
  PP_HIP_TRY(hipHostMalloc(p, cls));
  if (pool.enabled) { std::lock_guard<std::mutex> lock(pool.mu); pool.live_pinned[*p] = cls; }
  if (pool.poison) std::memset(*p, 0xFF, cls);
  return PP_OK;
}

void PoolPinnedFree(void* p) {
  if (!p) return;
  Pool& pool = P();
  {
    std::lock_guard<std::mutex> lock(pool.mu);
    auto it = pool.live_pinned.find(p);
    if (it != pool.live_pinned.end() && pool.enabled) {
      const size_t cls = it->second;
      pool.live_pinned.erase(it);
      if (pool.free_pinned[cls].size() < 64) { pool.free_pinned[cls].push_back(p); return; }
    } else if (it != pool.live_pinned.end()) pool.live_pinned.erase(it);
  }
  (void)hipHostFree(p);
}

int PoolStreamAcquire(hipStream_t* s) {
  Pool& pool = P();
  int device = 0;
  PP_HIP_TRY(hipGetDevice(&device));
  if (pool.enabled) {
    std::lock_guard<std::mutex> lock(pool.mu);
    auto& v = pool.streams[device];
    if (!v.empty()) { *s = v.back(); v.pop_back(); return PP_OK; }
  }
  PP_HIP_TRY(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
  if (pool.enabled) { std::lock_guard<std::mutex> lock(pool.mu); pool.stream_device[(void*)*s] = device; }
  return PP_OK;
}
void PoolStreamRelease(hipStream_t s) {
  if (!s) return;
  Pool& pool = P();
  {
    std::lock_guard<std::mutex> lock(pool.mu);
    auto it = pool.stream_device.find((void*)s);
    if (pool.enabled && it != pool.stream_device.end() && pool.streams[it->second].size() < 32) { pool.streams[it->second].push_back(s); return; }
    if (it != pool.stream_device.end()) pool.stream_device.erase(it);
  }
  (void)hipStreamDestroy(s);
}

int PoolEventAcquire(hipEvent_t* e, bool timing) {
  Pool& pool = P();
  int device = 0;
  PP_HIP_TRY(hipGetDevice(&device));
  if (pool.enabled) {
    std::lock_guard<std::mutex> lock(pool.mu);
    auto& v = pool.events[timing ? 1 : 0][device];
    if (!v.empty()) { *e = v.back(); v.pop_back(); return PP_OK; }
  }
  if (timing) PP_HIP_TRY(hipEventCreate(e));
  else PP_HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
  if (pool.enabled) { std::lock_guard<std::mutex> lock(pool.mu); pool.event_device[(void*)*e] = device; }
  return PP_OK;
}
void PoolEventRelease(hipEvent_t e, bool timing) {
  if (!e) return;
  Pool& pool = P();
  {
    std::lock_guard<std::mutex> lock(pool.mu);
    auto it = pool.event_device.find((void*)e);
    if (pool.enabled && it != pool.event_device.end() && pool.events[timing ? 1 : 0][it->second].size() < 512) { pool.events[timing ? 1 : 0][it->second].push_back(e); return; }
    if (it != pool.event_device.end()) pool.event_device.erase(it);
  }
  (void)hipEventDestroy(e);
}

void PoolTrim() {
  Pool& pool = P();
  std::vector<void*> dev, pinned;
  {
    std::lock_guard<std::mutex> lock(pool.mu);
    for (auto& kv : pool.free_dev) { dev.insert(dev.end(), kv.second.begin(), kv.second.end()); kv.second.clear(); }
    for (auto& kv : pool.free_pinned) { pinned.insert(pinned.end(), kv.second.begin(), kv.second.end()); kv.second.clear(); }
    pool.cached_bytes.clear();
  }
  for (void* p : dev) (void)hipFree(p);
  for (void* p : pinned) (void)hipHostFree(p);
}

}  // namespace ppsfm

extern "C" int pp_pool_trim(void) try { ppsfm::PoolTrim(); return PP_OK; } PP_API_CATCH("pp_pool_trim")

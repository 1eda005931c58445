// The Schur pair lists of pp_ba_create, built on the device.
//
// For every pair of variable images (ci >= cj) that share a variable point: the (observation of ci, observation of cj) pairs, lists in (ci, cj)
// order, a list's entries in (oi, oj) order (what k_schur_pairs / k_schur_blocks / the chunk kernels walk; csrc/ba_eval.hip builds the same lists on
// the host for small problems and as the fallback).  The mapper builds a new BundleAdjuster per global bundle adjustment
// (src/sfm/incremental_mapper.cc:893-936), and on the host these lists were the largest part of a create (4-8 ms at 200k observations: 1.6 M visits of the
// tracks for 0.7 M entries, memory-bound on one to eight host threads).  Here, with the by-point lists already uploaded:
//   k_pl_count    a thread per entry of the by-point lists: its point's other observers with an image <= its own -> one atomic add per entry into a
//                 C x C table of list lengths (low contention: the entries of one list come from different points)
//   scan          exclusive scan of the table = the lists' starts (three launches: per-span sums, their scan, the spans again)
//   k_pl_fill     the same walk: position = atomic add on the list's fill counter -> (oi, oj) written into the list - in arrival order
//   k_pl_rank     a thread per ENTRY: its rank among the entries of its list by (oi, oj) (the pairs of a list are distinct) = its place in the sorted list,
//                 written to a second buffer - which makes the result independent of the atomics' order (a thread per LIST with an insertion sort took 0.8 ms
//                 on a sequence scene's lists of ~35 entries; a structure with a list of more than kMaxSortedList entries goes back to the host builder)
//   k_pl_heads    non-empty lists flagged, scanned, compacted: pair_start / pair_ij in list order
// The entries stay on the device (they are what the kernels read); the host gets the 3 ints per list it needs for the tile map, the list order by length
// and strip, and the chunks.
#include <algorithm>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "ba_impl.hpp"
#include "resource_pool.hpp"

namespace ppsfm {
namespace {

constexpr int kScanThreads = 1024;
constexpr int kMaxSortedList = 2048;

// exclusive scan of n int32 (n <= kScanThreads * kScanThreads * span): per-span sums, their scan, the spans again
__device__ __forceinline__ int SpanLocalScan(int64_t i0, int64_t i1, const int32_t* __restrict__ in, int* part, int* local_out) {
  const int tid = threadIdx.x;
  const int64_t chunk = (i1 - i0 + kScanThreads - 1) / kScanThreads;
  const int64_t t0 = i0 + tid * chunk, t1 = (t0 + chunk < i1) ? t0 + chunk : i1;
  int local = 0;
  for (int64_t i = t0; i < t1; ++i) local += in[i];
  part[tid] = local;
  __syncthreads();
  for (int off = 1; off < kScanThreads; off <<= 1) {   // Hillis-Steele inclusive scan
    const int v = (tid >= off) ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  *local_out = local;
  return part[tid] - local;
}
__global__ __launch_bounds__(kScanThreads) void k_pl_scan_sums(int64_t n, int64_t span, const int32_t* __restrict__ in, int32_t* __restrict__ block_sums) {
  __shared__ int part[kScanThreads];
  const int64_t i0 = blockIdx.x * span, i1 = (i0 + span < n) ? i0 + span : n;
  int local;
  (void)SpanLocalScan(i0, i1 > i0 ? i1 : i0, in, part, &local);
  if (threadIdx.x == kScanThreads - 1) block_sums[blockIdx.x] = part[kScanThreads - 1];
}
__global__ __launch_bounds__(kScanThreads) void k_pl_scan_offsets(int nblocks, int32_t* __restrict__ block_sums, int32_t* __restrict__ total_out) {
  __shared__ int part[kScanThreads];
  const int tid = threadIdx.x;
  const int v0 = tid < nblocks ? block_sums[tid] : 0;
  part[tid] = v0;
  __syncthreads();
  for (int off = 1; off < kScanThreads; off <<= 1) {
    const int v = (tid >= off) ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  if (tid < nblocks) block_sums[tid] = part[tid] - v0;     // exclusive
  if (tid == kScanThreads - 1) *total_out = part[kScanThreads - 1];
}
__global__ __launch_bounds__(kScanThreads) void k_pl_scan_write(int64_t n, int64_t span, const int32_t* __restrict__ in, const int32_t* __restrict__ block_offsets,
                                                                int32_t* __restrict__ out) {
  __shared__ int part[kScanThreads];
  const int64_t i0 = blockIdx.x * span, i1r = (i0 + span < n) ? i0 + span : n, i1 = i1r > i0 ? i1r : i0;
  int local;
  int pos = block_offsets[blockIdx.x] + SpanLocalScan(i0, i1, in, part, &local);
  const int64_t chunk = (i1 - i0 + kScanThreads - 1) / kScanThreads;
  const int64_t t0 = i0 + threadIdx.x * chunk, t1 = (t0 + chunk < i1) ? t0 + chunk : i1;
  for (int64_t i = t0; i < t1; ++i) { const int v = in[i]; out[i] = pos; pos += v; }
}
struct ScanScratch { int32_t* block_sums; int32_t* total; };
void ExclusiveScan(const int32_t* in, int32_t* out, int64_t n, const ScanScratch& sc, hipStream_t s) {
  const int nblocks = (int)std::min<int64_t>(kScanThreads, (n + 4095) / 4096);
  const int64_t span = (n + nblocks - 1) / nblocks;
  hipLaunchKernelGGL(k_pl_scan_sums, dim3(nblocks), dim3(kScanThreads), 0, s, n, span, in, sc.block_sums);
  hipLaunchKernelGGL(k_pl_scan_offsets, dim3(1), dim3(kScanThreads), 0, s, nblocks, sc.block_sums, sc.total);
  hipLaunchKernelGGL(k_pl_scan_write, dim3(nblocks), dim3(kScanThreads), 0, s, n, span, in, (const int32_t*)sc.block_sums, out);
}

// the image of every entry of the by-point lists, -1 for a constant pose or a constant point (such entries take part in no list)
__global__ __launch_bounds__(256) void k_pl_images(int64_t M, const int32_t* __restrict__ pt_obs, const int32_t* __restrict__ obs_pose, const int32_t* __restrict__ obs_point,
                                                   const uint8_t* __restrict__ pose_const, const uint8_t* __restrict__ point_const, int32_t* __restrict__ pt_pose) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= M) return;
  const int o = pt_obs[e];
  const int c = obs_pose[o];
  pt_pose[e] = (pose_const[c] || point_const[obs_point[o]]) ? -1 : c;
}
// kFill = false: list lengths; true: the entries, in arrival order
template <bool kFill>
__global__ __launch_bounds__(256) void k_pl_walk(int64_t M, int C, const int32_t* __restrict__ pt_start, const int32_t* __restrict__ pt_obs, const int32_t* __restrict__ obs_point,
                                                 const int32_t* __restrict__ pt_pose, int32_t* __restrict__ table, const int32_t* __restrict__ list_start,
                                                 int32_t* __restrict__ entries, int32_t* __restrict__ entry_key) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= M) return;
  const int ci = pt_pose[e];
  if (ci < 0) return;
  const int32_t oi = pt_obs[e];
  const int p = obs_point[oi];
  const int f0 = pt_start[p], f1 = pt_start[p + 1];
  for (int f = f0; f < f1; ++f) {
    const int cj = pt_pose[f];
    if ((unsigned)cj > (unsigned)ci || f == (int)e) continue;      // constant (-1), a later image, or the (o,o) self term (k_schur_self's)
    const size_t key = (size_t)ci * C + cj;
    const int pos = atomicAdd(&table[key], 1);
    if (kFill) {
      const size_t at = (size_t)list_start[key] + pos;
      entries[2 * at] = oi; entries[2 * at + 1] = pt_obs[f];
      entry_key[at] = (int32_t)key;
    }
  }
}
// a thread per entry: its place in its list sorted by (oi, oj)
__global__ __launch_bounds__(256) void k_pl_rank(int64_t E, const int32_t* __restrict__ entry_key, const int32_t* __restrict__ count, const int32_t* __restrict__ list_start,
                                                 const long long* __restrict__ in, long long* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int key = entry_key[e];
  const int s = list_start[key], n = count[key];
  const long long v = in[e];      // (oi, oj) as one 64-bit word: low half oi
  const int vi = (int)(v & 0xffffffffll), vj = (int)(v >> 32);
  int rank = 0;
  for (int f = 0; f < n; ++f) {
    const long long u = in[s + f];
    const int ui = (int)(u & 0xffffffffll), uj = (int)(u >> 32);
    rank += (ui < vi || (ui == vi && uj < vj)) ? 1 : 0;
  }
  out[s + rank] = v;
}
__global__ __launch_bounds__(256) void k_pl_flags(int64_t num_keys, const int32_t* __restrict__ count, int32_t* __restrict__ flag) {
  const int64_t key = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (key < num_keys) flag[key] = count[key] > 0 ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_pl_heads(int64_t num_keys, int C, const int32_t* __restrict__ flag, const int32_t* __restrict__ rank, const int32_t* __restrict__ list_start,
                                                  int32_t* __restrict__ pair_start, int32_t* __restrict__ pair_ij) {
  const int64_t key = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (key >= num_keys || !flag[key]) return;
  const int r = rank[key];
  pair_start[r] = list_start[key];
  pair_ij[2 * r] = (int32_t)(key / C); pair_ij[2 * r + 1] = (int32_t)(key % C);
}
__global__ __launch_bounds__(256) void k_pl_max(int64_t n, const int32_t* __restrict__ v, int32_t* __restrict__ out) {
  int m = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = max(m, v[i]);
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// a wavefront per 64-bit word of the graph's bit matrix: bit j of row i = images i > j share a list
__global__ __launch_bounds__(256) void k_pl_bits(int C, int W, const int32_t* __restrict__ count, unsigned long long* __restrict__ bits) {
  const int64_t word = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (word >= (int64_t)C * W) return;
  const int i = (int)(word / W), w = (int)(word % W), j = 64 * w + (threadIdx.x & 63);
  const bool on = j < i && count[(size_t)i * C + j] > 0;
  const unsigned long long m = __ballot(on);
  if ((threadIdx.x & 63) == 0) bits[word] = m;
}

}  // namespace

bool PairListsOnDeviceEligible(int C, int64_t M) {
  if (const char* e = std::getenv("PPSFM_BA_PAIR_LISTS")) { if (e[0] == 'h' || e[0] == 'H') return false; if (e[0] == 'd' || e[0] == 'D') return C <= 2048; }
  return M >= 50000 && C <= 2048;      // (the C x C table: 16 MB at 2048 images; below 50k observations the host builder is as fast as the launches + the read-back)
}

int CoVisibilityOnDevice(int C, int64_t M, const int32_t* d_pt_start, const int32_t* d_pt_obs, const int32_t* d_obs_pose, const int32_t* d_obs_point,
                         const uint8_t* d_pose_const, const uint8_t* d_point_const, hipStream_t s, std::vector<uint64_t>* bits) {
  const int64_t K = (int64_t)C * C;
  const int W = (C + 63) / 64;
  void *q_pose = nullptr, *q_count = nullptr, *q_bits = nullptr;
  int rc;
  if ((rc = PoolDeviceAlloc(&q_pose, sizeof(int32_t) * (size_t)M)) || (rc = PoolDeviceAlloc(&q_count, sizeof(int32_t) * (size_t)K)) ||
      (rc = PoolDeviceAlloc(&q_bits, sizeof(uint64_t) * (size_t)C * W))) { PoolDeviceFree(q_pose); PoolDeviceFree(q_count); PoolDeviceFree(q_bits); return rc; }
  int32_t* pt_pose = (int32_t*)q_pose; int32_t* count = (int32_t*)q_count;
  bits->resize((size_t)C * W);
  const dim3 gm((unsigned)((M + 255) / 256));
  hipError_t e = hipMemsetAsync(count, 0, sizeof(int32_t) * K, s);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_pl_images, gm, dim3(256), 0, s, M, d_pt_obs, d_obs_pose, d_obs_point, d_pose_const, d_point_const, pt_pose);
    hipLaunchKernelGGL(k_pl_walk<false>, gm, dim3(256), 0, s, M, C, d_pt_start, d_pt_obs, d_obs_point, (const int32_t*)pt_pose, count, (const int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr);
    hipLaunchKernelGGL(k_pl_bits, dim3((unsigned)(((int64_t)C * W + 3) / 4)), dim3(256), 0, s, C, W, (const int32_t*)count, (unsigned long long*)q_bits);
    e = hipMemcpyAsync(bits->data(), q_bits, sizeof(uint64_t) * (size_t)C * W, hipMemcpyDeviceToHost, s);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e == hipSuccess) e = hipGetLastError();
  PoolDeviceFree(q_pose); PoolDeviceFree(q_count); PoolDeviceFree(q_bits);
  if (e != hipSuccess) { SetLastError("co-visibility on the device: %s", hipGetErrorString(e)); return PP_ERR_HIP; }
  return PP_OK;
}

// d_* : the problem's device arrays (internal image order).  On success: *entries_out (pool block: 2 x *num_entries ints, or null when there is none),
// pair_start (lists + 1) / pair_ij (2 x lists) on the host.  PP_ERR_INVALID with *fallback = true: a list longer than kMaxSortedList - the caller takes the host builder.
int BuildPairListsOnDevice(int C, int64_t M, const int32_t* d_pt_start, const int32_t* d_pt_obs, const int32_t* d_obs_pose, const int32_t* d_obs_point,
                           const uint8_t* d_pose_const, const uint8_t* d_point_const, hipStream_t s, int32_t** entries_out, int64_t* num_entries,
                           std::vector<int32_t>* pair_start, std::vector<int32_t>* pair_ij, bool* fallback) {
  *entries_out = nullptr; *num_entries = 0; *fallback = false;
  const int64_t K = (int64_t)C * C;
  void* blocks[12] = {nullptr};
  int nb = 0;
  auto alloc = [&](size_t ints, int32_t** p) { void* q = nullptr; const int rc = PoolDeviceAlloc(&q, ints * sizeof(int32_t)); if (!rc) { blocks[nb++] = q; *p = (int32_t*)q; } return rc; };
  auto release = [&]() { for (int i = 0; i < nb; ++i) PoolDeviceFree(blocks[i]); nb = 0; };
  int32_t *pt_pose, *count, *fill, *start, *flag, *small;
  int rc;
  if ((rc = alloc((size_t)M, &pt_pose)) || (rc = alloc((size_t)K, &count)) || (rc = alloc((size_t)K, &fill)) || (rc = alloc((size_t)K + 1, &start)) || (rc = alloc((size_t)K, &flag)) ||
      (rc = alloc(kScanThreads + 8, &small))) { release(); return rc; }
  ScanScratch sc{small, small + kScanThreads};
  int32_t* d_max = small + kScanThreads + 1;
  int32_t* d_lists = small + kScanThreads + 2;
  auto fail = [&](hipError_t e) { release(); SetLastError("pair lists on the device: %s", hipGetErrorString(e)); return PP_ERR_HIP; };
  hipError_t e;
  if ((e = hipMemsetAsync(count, 0, sizeof(int32_t) * K, s)) != hipSuccess) return fail(e);
  if ((e = hipMemsetAsync(fill, 0, sizeof(int32_t) * K, s)) != hipSuccess) return fail(e);
  if ((e = hipMemsetAsync(small, 0, sizeof(int32_t) * (kScanThreads + 8), s)) != hipSuccess) return fail(e);
  const dim3 gm((unsigned)((M + 255) / 256)), gk((unsigned)((K + 255) / 256));
  hipLaunchKernelGGL(k_pl_images, gm, dim3(256), 0, s, M, d_pt_obs, d_obs_pose, d_obs_point, d_pose_const, d_point_const, pt_pose);
  hipLaunchKernelGGL(k_pl_walk<false>, gm, dim3(256), 0, s, M, C, d_pt_start, d_pt_obs, d_obs_point, (const int32_t*)pt_pose, count, (const int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr);
  ExclusiveScan(count, start, K, sc, s);
  hipLaunchKernelGGL(k_pl_max, dim3(256), dim3(256), 0, s, K, (const int32_t*)count, d_max);
  int32_t host2[2] = {0, 0};      // total entries, longest list
  if ((e = hipMemcpyAsync(&host2[0], sc.total, sizeof(int32_t), hipMemcpyDeviceToHost, s)) != hipSuccess) return fail(e);
  if ((e = hipMemcpyAsync(&host2[1], d_max, sizeof(int32_t), hipMemcpyDeviceToHost, s)) != hipSuccess) return fail(e);
  if ((e = hipStreamSynchronize(s)) != hipSuccess) return fail(e);
  const int64_t E = host2[0];
  if (host2[1] > kMaxSortedList) { release(); *fallback = true; return PP_ERR_INVALID; }
  pair_start->assign(1, 0); pair_ij->clear();
  if (E == 0) { release(); return PP_OK; }
  int32_t* entries = nullptr;      // the sorted lists (the caller's); the lists in arrival order and the list of every entry: scratch
  int32_t *arrival = nullptr, *entry_key = nullptr;
  { void* q = nullptr; if ((rc = PoolDeviceAlloc(&q, sizeof(int32_t) * 2 * (size_t)E))) { release(); return rc; } entries = (int32_t*)q; }
  if ((rc = alloc(2 * (size_t)E, &arrival)) || (rc = alloc((size_t)E, &entry_key))) { PoolDeviceFree(entries); release(); return rc; }
  hipLaunchKernelGGL(k_pl_walk<true>, gm, dim3(256), 0, s, M, C, d_pt_start, d_pt_obs, d_obs_point, (const int32_t*)pt_pose, fill, (const int32_t*)start, arrival, entry_key);
  hipLaunchKernelGGL(k_pl_rank, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, E, (const int32_t*)entry_key, (const int32_t*)count, (const int32_t*)start,
                     (const long long*)arrival, (long long*)entries);
  hipLaunchKernelGGL(k_pl_flags, gk, dim3(256), 0, s, K, (const int32_t*)count, flag);
  // non-empty lists in key order: rank = exclusive scan of the flags (into `fill`, free again), their number in d_lists
  ScanScratch sc2{small, d_lists};
  ExclusiveScan(flag, fill, K, sc2, s);
  int32_t lists = 0;
  if ((e = hipMemcpyAsync(&lists, d_lists, sizeof(int32_t), hipMemcpyDeviceToHost, s)) != hipSuccess) { PoolDeviceFree(entries); return fail(e); }
  if ((e = hipStreamSynchronize(s)) != hipSuccess) { PoolDeviceFree(entries); return fail(e); }
  int32_t *d_pstart = nullptr, *d_pij = nullptr;
  if ((rc = alloc((size_t)lists + 1, &d_pstart)) || (rc = alloc(2 * (size_t)lists + 2, &d_pij))) { PoolDeviceFree(entries); release(); return rc; }
  hipLaunchKernelGGL(k_pl_heads, gk, dim3(256), 0, s, K, C, (const int32_t*)flag, (const int32_t*)fill, (const int32_t*)start, d_pstart, d_pij);
  pair_start->resize((size_t)lists + 1); pair_ij->resize(2 * (size_t)lists);
  if ((e = hipMemcpyAsync(pair_start->data(), d_pstart, sizeof(int32_t) * lists, hipMemcpyDeviceToHost, s)) != hipSuccess ||
      (e = hipMemcpyAsync(pair_ij->data(), d_pij, sizeof(int32_t) * 2 * lists, hipMemcpyDeviceToHost, s)) != hipSuccess ||
      (e = hipStreamSynchronize(s)) != hipSuccess || (e = hipGetLastError()) != hipSuccess) { PoolDeviceFree(entries); return fail(e); }
  (*pair_start)[(size_t)lists] = (int32_t)E;
  release();
  *entries_out = entries; *num_entries = E;
  return PP_OK;
}

// The same lists on the host: the builder of small problems, of structures with a list too long for the device's per-list sort, and the reference the device
// builder is tested against (bit for bit).  by-point CSR pt_start / pt_obs, obs_pose in the internal image order, list_const[c] != 0: image c has no columns,
// threads = 0: by size.  lap(what): pp_ba_create's profile.
void BuildPairListsOnHost(int C, int P, int64_t M, const int32_t* pt_start, const int32_t* pt_obs, const int32_t* obs_pose, const uint8_t* list_const,
                          const uint8_t* point_const, int threads, const std::function<void(const char*)>& lap, int64_t* total_entries_out,
                          std::vector<int32_t>* pair_start_out, std::vector<int32_t>* pair_ij_out, std::vector<int32_t>* pair_entries_out) {
  std::vector<int32_t>&pair_start = *pair_start_out, &pair_ij = *pair_ij_out, &pair_entries = *pair_entries_out;
  pair_start.clear(); pair_ij.clear(); pair_entries.clear();
  int64_t total_entries = 0;
  {
  // Point by point (sequential reads of the by-point lists), every entry dropped into the bucket of its ROW image ci - one append stream per image -, then every
  // row sorted by its column image with a counting sort over C cache-resident counters.  Three host threads-worth of independent pieces: points in ranges for
  // the two passes over the tracks, rows in ranges for the sort.  (Walking image by image instead - no buckets - gathers three cache lines per observation
  // and measured 5-8 ms at 200k observations; the C x C counter table of rounds 1-4 7.6 ms.)
  std::vector<int32_t> pt_pose(M);      // the image of every entry of the by-point lists (-1: a constant pose)
  for (int64_t e = 0; e < M; ++e) { const int c = obs_pose[pt_obs[e]]; pt_pose[e] = list_const[c] ? -1 : c; }
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const int nthreads = threads > 0 ? threads : (M >= 100000 ? (int)std::min<unsigned>(M >= 200000 ? 8u : 4u, hw) : 1);      // (the machine's usable cores may be fewer than it reports: a handful)
  auto parallel = [&](auto&& body) { ParallelFor(nthreads, body); };      // (common.hpp: a worker's exception reaches this thread, every thread is joined)
  std::vector<int32_t> pbeg(nthreads + 1, P);
  { pbeg[0] = 0; const int64_t per = (M + nthreads - 1) / nthreads; int t = 1; for (int p = 0; p < P && t < nthreads; ++p) if ((int64_t)pt_start[p + 1] >= per * t) pbeg[t++] = p + 1; }
  auto for_entries = [&](int p0, int p1, auto&& fn) {      // fn(ci, cj, oi, oj) for every entry of the points [p0, p1)
    for (int p = p0; p < p1; ++p) {
      if (point_const[p]) continue;
      const int e0 = pt_start[p], e1 = pt_start[p + 1];
      for (int e = e0; e < e1; ++e) {
        const int ci = pt_pose[e];
        if (ci < 0) continue;
        const int32_t oi = pt_obs[e];
        for (int f = e0; f < e1; ++f) {
          const int cj = pt_pose[f];
          if ((unsigned)cj > (unsigned)ci || f == e) continue;      // constant (-1), a later image, or the (o,o) self term (k_schur_self's)
          fn(ci, cj, oi, pt_obs[f]);
        }
      }
    }
  };
  lap("  pair lists: images of the by-point lists");
  // pass A: entries per (thread, row)
  std::vector<int32_t> tcount((size_t)nthreads * C, 0);
  parallel([&](int t) { int32_t* cnt = tcount.data() + (size_t)t * C; for_entries(pbeg[t], pbeg[t + 1], [&](int ci, int, int32_t, int32_t) { ++cnt[ci]; }); });
  std::vector<int64_t> row_off(C + 1, 0);
  for (int c = 0; c < C; ++c) { int64_t n = 0; for (int t = 0; t < nthreads; ++t) { const int32_t v = tcount[(size_t)t * C + c]; tcount[(size_t)t * C + c] = (int32_t)n; n += v; } row_off[c + 1] = row_off[c] + n; }
  total_entries = row_off[C];
  lap("  pair lists: pass A (counts)");
  // pass B: the buckets (column image, oi, oj), a row's entries in point order
  struct Raw { int32_t cj, oi, oj; };
  std::vector<Raw> raw((size_t)total_entries);
  parallel([&](int t) {
    std::vector<int64_t> at(C);
    for (int c = 0; c < C; ++c) at[c] = row_off[c] + tcount[(size_t)t * C + c];
    for_entries(pbeg[t], pbeg[t + 1], [&](int ci, int cj, int32_t oi, int32_t oj) { raw[(size_t)at[ci]++] = Raw{cj, oi, oj}; });
  });
  lap("  pair lists: pass B (buckets)");
  // the rows: lists in cj order, a list's entries in (oi, oj) order (the order of the walk when the observations are grouped by point with increasing
  // indices - BundleAdjuster::SetUp's order; sorted otherwise)
  pair_entries.resize(2 * (size_t)total_entries);
  std::vector<int32_t> cbeg(nthreads + 1, C);
  { cbeg[0] = 0; const int64_t per = (total_entries + nthreads - 1) / nthreads; int t = 1; for (int c = 0; c < C && t < nthreads; ++c) if (row_off[c + 1] >= per * t) cbeg[t++] = c + 1; }
  struct Lists { std::vector<int32_t> start, ij; };
  std::vector<Lists> lists(nthreads);
  parallel([&](int t) {
    Lists& o = lists[t];
    o.start.reserve((size_t)(row_off[cbeg[t + 1]] - row_off[cbeg[t]]) / 4 + 64); o.ij.reserve(o.start.capacity() * 2);
    std::vector<int32_t> cnt(C, 0), pos(C, 0), touched;
    for (int ci = cbeg[t]; ci < cbeg[t + 1]; ++ci) {
      const int64_t r0 = row_off[ci], r1 = row_off[ci + 1];
      if (r0 == r1) continue;
      touched.clear();
      for (int64_t q = r0; q < r1; ++q) if (cnt[raw[(size_t)q].cj]++ == 0) touched.push_back(raw[(size_t)q].cj);
      std::sort(touched.begin(), touched.end());
      int64_t at = r0;
      for (int cj : touched) { pos[cj] = (int32_t)at; o.start.push_back((int32_t)at); o.ij.push_back(ci); o.ij.push_back(cj); at += cnt[cj]; }
      for (int64_t q = r0; q < r1; ++q) { const Raw& e = raw[(size_t)q]; const size_t w = 2 * (size_t)pos[e.cj]++; pair_entries[w] = e.oi; pair_entries[w + 1] = e.oj; }
      for (int cj : touched) {
        const size_t l1 = (size_t)pos[cj], l0 = l1 - (size_t)cnt[cj];
        cnt[cj] = 0;
        bool sorted = true;
        for (size_t q = l0 + 1; q < l1 && sorted; ++q)
          sorted = pair_entries[2 * q - 2] < pair_entries[2 * q] || (pair_entries[2 * q - 2] == pair_entries[2 * q] && pair_entries[2 * q - 1] <= pair_entries[2 * q + 1]);
        if (!sorted) {
          int64_t* le = reinterpret_cast<int64_t*>(pair_entries.data() + 2 * l0);      // (oi, oj) pairs as they lie: sorted as pairs
          std::vector<std::pair<int32_t, int32_t>> tmp(l1 - l0);
          for (size_t q = l0; q < l1; ++q) tmp[q - l0] = {pair_entries[2 * q], pair_entries[2 * q + 1]};
          std::sort(tmp.begin(), tmp.end());
          for (size_t q = l0; q < l1; ++q) { pair_entries[2 * q] = tmp[q - l0].first; pair_entries[2 * q + 1] = tmp[q - l0].second; }
          (void)le;
        }
      }
    }
  });
  lap("  pair lists: rows sorted");
  size_t nl = 0;
  for (const Lists& o : lists) nl += o.start.size();
  pair_start.reserve(nl + 1); pair_ij.reserve(2 * nl);
  for (const Lists& o : lists) { pair_start.insert(pair_start.end(), o.start.begin(), o.start.end()); pair_ij.insert(pair_ij.end(), o.ij.begin(), o.ij.end()); }
  pair_start.push_back((int32_t)total_entries);
  }
  *total_entries_out = total_entries;
}

}  // namespace ppsfm

// Host-only entry point of the builder above (tests, the sanitizer build): the lists of a problem description in the CALLER's image order.
extern "C" int pp_ba_pair_lists_host(const pp_ba_problem_desc* d, int32_t threads, int64_t* num_lists, int64_t* num_entries, int32_t* pair_start, int32_t* pair_ij,
                                     int32_t* pair_entries, int64_t capacity_lists, int64_t capacity_entries) try {
  using namespace ppsfm;
  PP_REQUIRE(d && num_lists && num_entries && d->obs_pose && d->obs_point && threads >= 0 && threads <= 64, "pp_ba_pair_lists_host: bad argument");
  const int C = d->num_poses, P = d->num_points;
  const int64_t M = d->num_obs;
  PP_REQUIRE(C > 0 && P > 0 && M >= 0 && M < ((int64_t)1 << 31), "pp_ba_pair_lists_host: empty or oversized problem");
  for (int64_t o = 0; o < M; ++o)
    PP_REQUIRE(d->obs_pose[o] >= 0 && d->obs_pose[o] < C && d->obs_point[o] >= 0 && d->obs_point[o] < P, "pp_ba_pair_lists_host: observation %lld indexes out of range", (long long)o);
  std::vector<int32_t> pt_start(P + 1, 0), pt_obs(M);      // CSR by point (counting sort keeps observation order inside a group), as pp_ba_create
  for (int64_t o = 0; o < M; ++o) pt_start[d->obs_point[o] + 1]++;
  for (int p = 0; p < P; ++p) pt_start[p + 1] += pt_start[p];
  { std::vector<int32_t> f(pt_start.begin(), pt_start.end() - 1); for (int64_t o = 0; o < M; ++o) pt_obs[f[d->obs_point[o]]++] = (int32_t)o; }
  std::vector<uint8_t> list_const(C, 0), point_const(P, 0);
  if (d->pose_const) std::memcpy(list_const.data(), d->pose_const, C);
  if (d->point_const) std::memcpy(point_const.data(), d->point_const, P);
  int64_t bound = 0;      // (pp_ba_create's 32-bit check)
  for (int p = 0; p < P; ++p) {
    if (point_const[p]) continue;
    int64_t nv = 0;
    for (int e = pt_start[p]; e < pt_start[p + 1]; ++e) nv += list_const[d->obs_pose[pt_obs[e]]] ? 0 : 1;
    bound += nv * (nv - 1);
  }
  PP_REQUIRE(bound < ((int64_t)1 << 31) - 1, "pp_ba_pair_lists_host: %lld Schur pair entries exceed the 32-bit pair lists", (long long)bound);
  std::vector<int32_t> ps, pij, pe;
  int64_t total = 0;
  BuildPairListsOnHost(C, P, M, pt_start.data(), pt_obs.data(), d->obs_pose, list_const.data(), point_const.data(), threads, [](const char*) {}, &total, &ps, &pij, &pe);
  *num_lists = (int64_t)ps.size() - 1; *num_entries = total;
  if (pair_start) std::memcpy(pair_start, ps.data(), sizeof(int32_t) * (size_t)std::min<int64_t>((int64_t)ps.size(), capacity_lists + 1));
  if (pair_ij) std::memcpy(pair_ij, pij.data(), sizeof(int32_t) * 2 * (size_t)std::min<int64_t>(*num_lists, capacity_lists));
  if (pair_entries) std::memcpy(pair_entries, pe.data(), sizeof(int32_t) * 2 * (size_t)std::min<int64_t>(total, capacity_entries));
  return PP_OK;
} PP_API_CATCH("pp_ba_pair_lists_host")

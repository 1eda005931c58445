// RCCL communicator of a point-sharded bundle adjustment group ("one BA across k GPUs").
//
// The reference has no multi-GPU path; the exchange it would need is the sum of the per-shard normal equations once per
// LM iteration.  One process per GPU; each process creates a pp_comm over the ranks of ITS sub-model's group (BASELINE
// configs[4]: 4 sub-models over 8 GPUs = 4 groups of 2) and hands it to its pp_ba_handle (pp_ba_set_communicator).  The
// solver then issues its reductions as RCCL collectives ON THE HANDLE'S STREAM - no host synchronisation, the speculative
// LM loop stays as it is, and every rank takes the same decisions because every rank reads the same reduced scalars.
//
// librccl is loaded with dlopen at the first use (pp_comm_unique_id / pp_comm_create), so libppsfm_hip.so itself has no
// link-time dependency on it: a single-GPU host without RCCL loads and runs the library unchanged.
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "common.hpp"
#include "rccl_comm.hpp"

namespace ppsfm {
namespace {
// the part of <rccl/rccl.h> used here (ABI-stable NCCL signatures)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { kNcclSuccess = 0, kNcclSum = 0, kNcclMax = 2, kNcclFloat64 = 8 };
struct Api {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  void* lib = nullptr;
  bool ok = false;
};
Api g_api;
std::once_flag g_once;

void LoadApi() {
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names) {
    g_api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_api.lib) break;
  }
  if (!g_api.lib) return;
#define SYM(field, name) g_api.field = reinterpret_cast<decltype(g_api.field)>(dlsym(g_api.lib, name))
  SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
  SYM(AllReduce, "ncclAllReduce"); SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_api.ok = g_api.GetUniqueId && g_api.CommInitRank && g_api.CommDestroy && g_api.AllReduce && g_api.GroupStart && g_api.GroupEnd;
}
int RequireApi() {
  std::call_once(g_once, LoadApi);
  if (!g_api.ok) { SetLastError("librccl.so could not be loaded (%s)", g_api.lib ? "missing symbols" : dlerror()); return PP_ERR_HIP; }
  return PP_OK;
}
int Check(ncclResult_t r, const char* what) {
  if (r == kNcclSuccess) return PP_OK;
  SetLastError("%s failed: %s", what, g_api.GetErrorString ? g_api.GetErrorString(r) : "rccl error");
  return PP_ERR_HIP;
}
}  // namespace

int CommAllReduce(pp_comm_impl* c, double* ptr, int64_t count, int op, hipStream_t s) {
  if (count <= 0) return PP_OK;
  return Check(g_api.AllReduce(ptr, ptr, (size_t)count, kNcclFloat64, op == PP_REDUCE_MAX ? kNcclMax : kNcclSum, reinterpret_cast<ncclComm_t>(c->comm), s), "ncclAllReduce");
}
int CommGroupStart() { return Check(g_api.GroupStart(), "ncclGroupStart"); }
int CommGroupEnd() { return Check(g_api.GroupEnd(), "ncclGroupEnd"); }
}  // namespace ppsfm

using namespace ppsfm;

extern "C" {

int pp_comm_unique_id(uint8_t* id) try {
  PP_REQUIRE(id, "pp_comm_unique_id: null");
  int rc = RequireApi(); if (rc) return rc;
  ncclUniqueId u;
  if ((rc = Check(g_api.GetUniqueId(&u), "ncclGetUniqueId"))) return rc;
  std::memcpy(id, u.internal, PP_COMM_ID_BYTES);
  return PP_OK;
} PP_API_CATCH("pp_comm_unique_id")

int pp_comm_create(const uint8_t* id, int32_t num_ranks, int32_t rank, int device, pp_comm_handle* out) try {
  PP_REQUIRE(id && out && num_ranks >= 1 && rank >= 0 && rank < num_ranks, "pp_comm_create: bad argument");
  *out = nullptr;
  int rc = RequireApi(); if (rc) return rc;
  PP_HIP_TRY(hipSetDevice(device));
  ncclUniqueId u;
  std::memcpy(u.internal, id, PP_COMM_ID_BYTES);
  pp_comm_impl* c = new pp_comm_impl();
  c->device = device; c->rank = rank; c->size = num_ranks;
  if ((rc = Check(g_api.CommInitRank(reinterpret_cast<ncclComm_t*>(&c->comm), num_ranks, u, rank), "ncclCommInitRank"))) { delete c; return rc; }
  *out = c;
  return PP_OK;
} PP_API_CATCH("pp_comm_create")

int pp_comm_destroy(pp_comm_handle c) try {
  if (!c) return PP_OK;
  if (c->comm && g_api.ok) (void)g_api.CommDestroy(reinterpret_cast<ncclComm_t>(c->comm));
  delete c;
  return PP_OK;
} PP_API_CATCH("pp_comm_destroy")

int pp_comm_allreduce(pp_comm_handle c, double* device_ptr, int64_t count, int32_t op) try {
  PP_REQUIRE(c && device_ptr && count >= 0, "pp_comm_allreduce: bad argument");
  PP_HIP_TRY(hipSetDevice(c->device));
  int rc = CommAllReduce(c, device_ptr, count, op, nullptr); if (rc) return rc;
  PP_HIP_TRY(hipStreamSynchronize(nullptr));
  return PP_OK;
} PP_API_CATCH("pp_comm_allreduce")

}  // extern "C"

// Internal face of the RCCL communicator (rccl_comm.hip) for the bundle-adjustment solver.
#pragma once
#include "common.hpp"

struct pp_comm_impl {
  void* comm = nullptr;      // ncclComm_t
  int device = 0;
  int32_t rank = 0, size = 1;
};

namespace ppsfm {
// in-place all-reduce of `count` doubles on stream s (stream-ordered, no host synchronisation); op = PP_REDUCE_*
int CommAllReduce(pp_comm_impl* c, double* ptr, int64_t count, int op, hipStream_t s);
// several CommAllReduce calls between the two become one RCCL launch
int CommGroupStart();
int CommGroupEnd();
}  // namespace ppsfm

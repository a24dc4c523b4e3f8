// Run-time binding of the handful of RCCL entry points the stripe exchange uses (api_stripes.cpp:
// uhdr_hip_comm_*, uhdr_hip_generate_gainmap_striped_dev).  libuhdr_hip.so does not link against librccl: a process
// that already carries a copy (PyTorch ships its own) keeps using that one, otherwise librccl.so.1 is loaded on the
// first communicator call; single-GPU users never load it.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only

namespace uhdr {

struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
const RcclApi& rccl();

}  // namespace uhdr

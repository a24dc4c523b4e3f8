// See rccl_bind.h.  The only dynamic loading in the product library, and the only library it may name is RCCL
// (tests/test_abi.py enforces both).
#include "rccl_bind.h"

#include <dlfcn.h>

namespace uhdr {

const RcclApi& rccl() {
  static const RcclApi api = [] {
    RcclApi a;
    void* h = nullptr;
    if (dlsym(RTLD_DEFAULT, "ncclAllReduce") == nullptr) {  // not in the process yet
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
      }
    }
    auto sym = [&](const char* n) -> void* {
      void* p = h ? dlsym(h, n) : nullptr;
      return p ? p : dlsym(RTLD_DEFAULT, n);
    };
    a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
    a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
    a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
    a.Send = (decltype(a.Send))sym("ncclSend");
    a.Recv = (decltype(a.Recv))sym("ncclRecv");
    a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.CommCount = (decltype(a.CommCount))sym("ncclCommCount");
    a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
    a.ok = a.GetUniqueId && a.CommInitRank && a.AllReduce && a.AllGather && a.Send && a.Recv && a.GroupStart && a.GroupEnd && a.CommDestroy && a.CommCount &&
           a.GetErrorString;
    return a;
  }();
  return api;
}

}  // namespace uhdr

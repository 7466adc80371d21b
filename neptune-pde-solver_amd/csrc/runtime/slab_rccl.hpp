// slab_rccl.hpp -- the multi-GPU face of one neptune_ir.apply, in C: dim-0 slab halo exchange over RCCL
// (ncclSend / ncclRecv grouped on a communication stream) overlapped with the interior update.
//
// The reference has no domain decomposition (every PETSc object lives on PETSC_COMM_SELF,
// lib/Runtime/PETSc/NeptunePETScRuntime.cpp:136,244,257); SURVEY.md 8(e) defines this path: rank g owns planes
// [start_g, stop_g) of dim 0 and holds `radius` ghost planes per existing neighbour in the same dense buffer, so a
// halo is one contiguous run of planes, sent and received in place.
//
// RCCL is bound at run time (dlopen of librccl.so.1, reusing the copy a host program such as PyTorch has already
// loaded): libneptune_hip.so itself has no link-time dependency on it, and single-GPU users never load it.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "../../../include/neptune_hip.h"

namespace neptune_hip {
namespace slab {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  char error[256] = "";
};

// nullptr (and api.error set) if RCCL cannot be loaded
inline RcclApi* rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)  // a copy already in the process (PyTorch ships one under the same soname) wins
      if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    for (const char* n : names)
      if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!api.handle) {
      snprintf(api.error, sizeof api.error, "cannot load librccl.so.1: %s", dlerror());
      return;
    }
    auto sym = [&](const char* name) {
      void* p = dlsym(api.handle, name);
      if (!p && !api.error[0]) snprintf(api.error, sizeof api.error, "librccl has no %s", name);
      return p;
    };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
  });
  return (api.handle && !api.error[0]) ? &api : nullptr;
}

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  char error[320] = "";   // last failure, for neptune_hip_slab_last_error
};

inline thread_local char g_last_error[320] = "";
inline void set_error(Comm* c, const char* what, const char* detail) {
  snprintf(g_last_error, sizeof g_last_error, "%s: %s", what, detail ? detail : "?");
  if (c) memcpy(c->error, g_last_error, sizeof c->error);
}

#define NEPTUNE_RCCL_TRY(c, api, call)                                  \
  do {                                                                  \
    const ncclResult_t _r = (call);                                     \
    if (_r != ncclSuccess) {                                            \
      ::neptune_hip::slab::set_error((c), #call, (api)->GetErrorString ? (api)->GetErrorString(_r) : "rccl error"); \
      return NEPTUNE_HIP_ECOMM;                                         \
    }                                                                   \
  } while (0)
#define NEPTUNE_HIP_TRY(c, call)                                        \
  do {                                                                  \
    const hipError_t _e = (call);                                       \
    if (_e != hipSuccess) {                                             \
      ::neptune_hip::slab::set_error((c), #call, hipGetErrorString(_e));                     \
      return NEPTUNE_HIP_ECOMM;                                         \
    }                                                                   \
  } while (0)

// ghost planes of ONE dense local buffer [r_lo ghost | n_own owned | r_hi ghost] x plane_bytes, in place
inline int exchange(Comm* c, void* field, size_t plane_bytes, int64_t n_own, int r_lo, int r_hi, int peer_lo,
                    int peer_hi, hipStream_t stream) {
  if (!c || !field || n_own <= 0 || r_lo < 0 || r_hi < 0) return NEPTUNE_HIP_EINVAL;
  if ((r_lo > 0 && (peer_lo < 0 || peer_lo >= c->world)) || (r_hi > 0 && (peer_hi < 0 || peer_hi >= c->world)))
    return NEPTUNE_HIP_EINVAL;
  if (r_lo > n_own || r_hi > n_own) return NEPTUNE_HIP_EINVAL;  // a halo deeper than the slab would need two hops
  if (r_lo == 0 && r_hi == 0) return NEPTUNE_HIP_OK;
  RcclApi* api = rccl();
  if (!api) { set_error(c, "rccl", rccl() ? "" : "library not available"); return NEPTUNE_HIP_ECOMM; }
  char* base = static_cast<char*>(field);
  char* own = base + (size_t)r_lo * plane_bytes;
  NEPTUNE_RCCL_TRY(c, api, api->GroupStart());
  // the neighbour below needs my first r_lo owned planes as ITS upper ghosts, and sends its last ones for my lower
  // ghosts; symmetric radius on both sides of a cut (both ranks run the same stencil)
  if (r_lo > 0) {
    NEPTUNE_RCCL_TRY(c, api, api->Send(own, (size_t)r_lo * plane_bytes, ncclUint8, peer_lo, c->comm, stream));
    NEPTUNE_RCCL_TRY(c, api, api->Recv(base, (size_t)r_lo * plane_bytes, ncclUint8, peer_lo, c->comm, stream));
  }
  if (r_hi > 0) {
    char* last = own + (size_t)(n_own - r_hi) * plane_bytes;
    NEPTUNE_RCCL_TRY(c, api, api->Send(last, (size_t)r_hi * plane_bytes, ncclUint8, peer_hi, c->comm, stream));
    NEPTUNE_RCCL_TRY(c, api, api->Recv(own + (size_t)n_own * plane_bytes, (size_t)r_hi * plane_bytes, ncclUint8, peer_hi,
                                       c->comm, stream));
  }
  NEPTUNE_RCCL_TRY(c, api, api->GroupEnd());
  return NEPTUNE_HIP_OK;
}

// One sharded apply: everything decided once, a step is then a fixed sequence of stream operations.
struct Plan {
  Comm* comm = nullptr;
  neptune_hip_apply_fn fn = nullptr;   // a lowered apply's geometry-level entry, or nullptr = built-in `body`
  int body = -1;
  int num_inputs = 1, esize = 8;
  int r_lo = 0, r_hi = 0, peer_lo = -1, peer_hi = -1;
  int64_t n_own = 0;
  size_t plane_bytes[NEPTUNE_HIP_MAX_INPUTS] = {0, 0, 0, 0};
  bool has_interior = false;
  int n_edges = 0;
  neptune_hip_apply_geom_t whole{}, interior{}, edges[2]{};
  neptune_hip_launch_cfg_t cfg{0, -1, 0, 0};
  bool has_cfg = false;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ready = nullptr, halo_done = nullptr;
};

}  // namespace slab
}  // namespace neptune_hip

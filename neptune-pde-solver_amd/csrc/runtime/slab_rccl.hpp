// slab_rccl.hpp -- the multi-GPU face of one neptune_ir.apply, in C: dim-0 slab halo exchange overlapped with the
// interior update.  Two transports behind one communicator type: RCCL (ncclSend / ncclRecv grouped on a communication
// stream) and peer copies (slab_peer.hpp: hipMemcpyAsync into the neighbour's mapped ghost planes, no CU moves data).
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
#include <sched.h>

#include "../../../include/neptune_hip.h"
#include "slab_peer.hpp"

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
  int transport = NEPTUNE_HIP_TRANSPORT_RCCL;
  ncclComm_t comm = nullptr;        // RCCL transport
  peer::State* peer = nullptr;      // peer-copy transport
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

// ---- peer-copy transport: the two handshake kernels are defined in slab_rccl.hip ---------------------------------
// store `v_lo` / `v_hi` to the (non-null) signal words, then wait until the (non-null) wait words reach the same values
void peer_signal_wait(hipStream_t stream, uint64_t* sig_lo, uint64_t* sig_hi, const uint64_t* wait_lo, const uint64_t* wait_hi,
                      uint64_t v_lo, uint64_t v_hi, uint32_t* err, uint32_t code);

inline int peer_exchange_many(Comm* c, void* const* fields, const size_t* plane_bytes, int nfields, int64_t n_own, int r_lo, int r_hi,
                              int peer_lo, int peer_hi, hipStream_t stream) {
  peer::State* s = c->peer;
  auto fail = [&](const char* what, const char* detail) { set_error(c, what, detail); return NEPTUNE_HIP_ECOMM; };
  if (*s->err_host) {
    char msg[160];
    snprintf(msg, sizeof msg, "a device-side wait of an earlier exchange timed out (code %u: %s); the ghost planes since then are stale",
             *s->err_host, (*s->err_host & 1) ? "neighbour never freed its ghost planes" : "neighbour's planes never arrived");
    return fail("peer transport", msg);
  }
  const int peers[2] = {peer_lo, peer_hi}, r[2] = {r_lo, r_hi};
  peer::RankShm& me = s->shm->ranks[s->rank];
  uint64_t n[2] = {0, 0};
  // 1. publish where my ghost planes are, per side and field (field f of exchange n uses ring entry (n + f) -- a call
  //    with several fields takes one exchange number per field so that both sides count alike)
  for (int side = 0; side < 2; ++side) {
    if (r[side] == 0) continue;
    for (int f = 0; f < nfields; ++f) {
      const uint64_t seq = s->seq[side] + 1 + (uint64_t)f;
      char* base = static_cast<char*>(fields[f]);
      char* ghost = side == peer::LO ? base : base + (size_t)(r_lo + n_own) * plane_bytes[f];
      uint32_t w = 0;
      uint64_t off = 0;
      // the whole local buffer must lie in one allocation; registering it by its first byte covers both ghost runs
      if (peer::window_of(s, base, (size_t)(r_lo + n_own + r_hi) * plane_bytes[f], &w, &off) != 0) return fail("peer transport", s->error);
      peer::Target& t = me.target[side][seq % peer::kRing];
      t.window = w;
      t.offset = off + (uint64_t)(ghost - base);
      t.bytes = (uint64_t)r[side] * plane_bytes[f];
      t.seq.store(seq, std::memory_order_release);
    }
    n[side] = s->seq[side] + (uint64_t)nfields;
  }
  // 2. K1: my ghost planes are free for these pushes (all earlier work on `stream` is complete); wait for the neighbours' word
  peer::Shm* d = s->dshm;
  auto mail = [&](int rank_) -> peer::RankShm* { return &d->ranks[rank_]; };
  uint64_t* sig_free[2] = {nullptr, nullptr};
  uint64_t* sig_arrived[2] = {nullptr, nullptr};
  const uint64_t* wait_free[2] = {nullptr, nullptr};
  const uint64_t* wait_arrived[2] = {nullptr, nullptr};
  // which side of the NEIGHBOUR my pushes land on: its opposite side -- except in a one-sided loop-back (an emulated edge
  // rank: the neighbour is this rank itself, which has no ghost planes on the opposite side), where the planes come back
  // into my own ghost planes of the same side, as an RCCL loop-back would deliver them
  auto landing = [&](int side) { return (peers[side] == s->rank && r[1 - side] == 0) ? side : 1 - side; };
  for (int side = 0; side < 2; ++side) {
    if (r[side] == 0) continue;
    const int opp = landing(side);
    sig_free[side] = &mail(peers[side])->peer_free[opp];      // I am the neighbour's `opp`-side neighbour
    sig_arrived[side] = &mail(peers[side])->arrived[opp];
    wait_free[side] = &mail(s->rank)->peer_free[side];
    wait_arrived[side] = &mail(s->rank)->arrived[side];
  }
  peer_signal_wait(stream, sig_free[0], sig_free[1], wait_free[0], wait_free[1], n[0], n[1], s->err_dev, 1);
  // 3. the pushes: my edge planes -> the neighbour's ghost planes (its descriptor exists once it has called this exchange)
  for (int side = 0; side < 2; ++side) {
    if (r[side] == 0) continue;
    const int opp = landing(side);
    peer::RankShm& nb = s->shm->ranks[peers[side]];
    for (int f = 0; f < nfields; ++f) {
      const uint64_t seq = s->seq[side] + 1 + (uint64_t)f;
      peer::Target& t = nb.target[opp][seq % peer::kRing];
      const double t0 = peer::now_s();
      while (t.seq.load(std::memory_order_acquire) != seq) {
        if (t.seq.load(std::memory_order_acquire) > seq) return fail("peer transport", "neighbour is ahead by more than the descriptor ring (ranks disagree on the number of exchanges)");
        if (peer::now_s() - t0 > peer::timeout_s()) {
          char msg[160];
          snprintf(msg, sizeof msg, "rank %d never reached exchange %llu on its %s side (it is at %llu)", peers[side], (unsigned long long)seq,
                   opp == peer::LO ? "lower" : "upper", (unsigned long long)t.seq.load());
          return fail("peer transport", msg);
        }
        sched_yield();
      }
      const size_t bytes = (size_t)r[side] * plane_bytes[f];
      if (t.bytes != bytes) return fail("peer transport", "neighbours disagree on the size of a halo");
      char* remote = peer::map_window(s, peers[side], t.window);
      if (!remote) return fail("peer transport", s->error);
      char* base = static_cast<char*>(fields[f]);
      char* own = base + (size_t)r_lo * plane_bytes[f];
      const char* src = side == peer::LO ? own : own + (size_t)(n_own - r_hi) * plane_bytes[f];
      NEPTUNE_HIP_TRY(c, hipMemcpyAsync(remote + t.offset, src, bytes, hipMemcpyDeviceToDevice, stream));
    }
  }
  // 4. K2: tell the neighbours their planes have landed; wait for mine
  peer_signal_wait(stream, sig_arrived[0], sig_arrived[1], wait_arrived[0], wait_arrived[1], n[0], n[1], s->err_dev, 2);
  for (int side = 0; side < 2; ++side)
    if (r[side] > 0) s->seq[side] = n[side];
  return NEPTUNE_HIP_OK;
}

// ghost planes of `nfields` dense local buffers [r_lo ghost | n_own owned | r_hi ghost] x plane_bytes[f], in place
inline int exchange_many(Comm* c, void* const* fields, const size_t* plane_bytes, int nfields, int64_t n_own, int r_lo, int r_hi,
                         int peer_lo, int peer_hi, hipStream_t stream) {
  if (!c || !fields || nfields < 1 || n_own <= 0 || r_lo < 0 || r_hi < 0) return NEPTUNE_HIP_EINVAL;
  for (int f = 0; f < nfields; ++f)
    if (!fields[f]) return NEPTUNE_HIP_EINVAL;
  if ((r_lo > 0 && (peer_lo < 0 || peer_lo >= c->world)) || (r_hi > 0 && (peer_hi < 0 || peer_hi >= c->world)))
    return NEPTUNE_HIP_EINVAL;
  if (r_lo > n_own || r_hi > n_own) return NEPTUNE_HIP_EINVAL;  // a halo deeper than the slab would need two hops
  if (r_lo == 0 && r_hi == 0) return NEPTUNE_HIP_OK;
  if (c->transport == NEPTUNE_HIP_TRANSPORT_PEER)
    return peer_exchange_many(c, fields, plane_bytes, nfields, n_own, r_lo, r_hi, peer_lo, peer_hi, stream);
  RcclApi* api = rccl();
  if (!api) { set_error(c, "rccl", rccl() ? "" : "library not available"); return NEPTUNE_HIP_ECOMM; }
  NEPTUNE_RCCL_TRY(c, api, api->GroupStart());
  for (int f = 0; f < nfields; ++f) {
    char* base = static_cast<char*>(fields[f]);
    char* own = base + (size_t)r_lo * plane_bytes[f];
    // the neighbour below needs my first r_lo owned planes as ITS upper ghosts, and sends its last ones for my lower
    // ghosts; symmetric radius on both sides of a cut (both ranks run the same stencil)
    if (r_lo > 0) {
      NEPTUNE_RCCL_TRY(c, api, api->Send(own, (size_t)r_lo * plane_bytes[f], ncclUint8, peer_lo, c->comm, stream));
      NEPTUNE_RCCL_TRY(c, api, api->Recv(base, (size_t)r_lo * plane_bytes[f], ncclUint8, peer_lo, c->comm, stream));
    }
    if (r_hi > 0) {
      char* last = own + (size_t)(n_own - r_hi) * plane_bytes[f];
      NEPTUNE_RCCL_TRY(c, api, api->Send(last, (size_t)r_hi * plane_bytes[f], ncclUint8, peer_hi, c->comm, stream));
      NEPTUNE_RCCL_TRY(c, api, api->Recv(own + (size_t)n_own * plane_bytes[f], (size_t)r_hi * plane_bytes[f], ncclUint8, peer_hi,
                                         c->comm, stream));
    }
  }
  NEPTUNE_RCCL_TRY(c, api, api->GroupEnd());
  return NEPTUNE_HIP_OK;
}
inline int exchange(Comm* c, void* field, size_t plane_bytes, int64_t n_own, int r_lo, int r_hi, int peer_lo,
                    int peer_hi, hipStream_t stream) {
  return exchange_many(c, &field, &plane_bytes, 1, n_own, r_lo, r_hi, peer_lo, peer_hi, stream);
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
  hipStream_t comm_stream = nullptr;   // created with the greatest priority: the exchange is dispatched ahead of the CU-filling interior grid
  hipEvent_t ready = nullptr, halo_done = nullptr, edges_done = nullptr;
  // optional per-step timing (neptune_hip_slab_plan_timing): a ring of timed events, read back without a sync per step
  static constexpr int kTimingRing = 64;
  struct StepEvents { hipEvent_t x0, x1, i0, i1, e1; };
  bool timing = false;
  int64_t timed_steps = 0;
  StepEvents* tev = nullptr;           // kTimingRing sets, created when timing is first switched on
};

// NEPTUNE_HIP_EDGES=compute keeps the edge launches of a sharded step on the compute stream, behind the interior launch (the
// schedule of rounds 1-2; measurements).  Default: on the communication stream, behind the exchange (neptune_hip_slab_apply).
inline bool edges_on_compute_stream() {
  static const bool v = [] { const char* e = getenv("NEPTUNE_HIP_EDGES"); return e && !strcmp(e, "compute"); }();
  return v;
}

}  // namespace slab
}  // namespace neptune_hip

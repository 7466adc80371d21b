// slab_rccl.hip -- include/neptune_hip.h section 8: halo exchange over RCCL and the sharded apply plan (host code only;
// its own translation unit so that it builds in seconds, linked into libneptune_hip.so).
#include "slab_rccl.hpp"

#include "../kernels/apply_launch.hpp"   // geom_validate (no kernel is instantiated here)

using namespace neptune_hip;

extern "C" {

// ---------------------------------------------------------------- slab decomposition over RCCL
struct neptune_hip_slab_comm : neptune_hip::slab::Comm {};
struct neptune_hip_slab_plan : neptune_hip::slab::Plan {};

int neptune_hip_slab_unique_id(void* id_out) {
  if (!id_out) return NEPTUNE_HIP_EINVAL;
  slab::RcclApi* api = slab::rccl();
  if (!api) { slab::set_error(nullptr, "rccl", "librccl.so.1 is not loadable"); return NEPTUNE_HIP_ECOMM; }
  ncclUniqueId id;
  static_assert(sizeof(id) == NEPTUNE_HIP_SLAB_ID_BYTES, "ncclUniqueId size");
  NEPTUNE_RCCL_TRY(nullptr, api, api->GetUniqueId(&id));
  memcpy(id_out, &id, sizeof id);
  return NEPTUNE_HIP_OK;
}

neptune_hip_slab_comm_t* neptune_hip_slab_comm_create(const void* id, int rank, int world) {
  if (world < 1 || rank < 0 || rank >= world || (!id && world > 1)) {
    slab::set_error(nullptr, "neptune_hip_slab_comm_create", "bad rank / world / id");
    return nullptr;
  }
  slab::RcclApi* api = slab::rccl();
  if (!api) { slab::set_error(nullptr, "rccl", "librccl.so.1 is not loadable"); return nullptr; }
  ncclUniqueId uid;
  if (id) memcpy(&uid, id, sizeof uid);
  else if (api->GetUniqueId(&uid) != ncclSuccess) { slab::set_error(nullptr, "ncclGetUniqueId", "failed"); return nullptr; }
  auto* c = new neptune_hip_slab_comm;
  c->rank = rank;
  c->world = world;
  const ncclResult_t r = api->CommInitRank(&c->comm, world, uid, rank);
  if (r != ncclSuccess) {
    slab::set_error(nullptr, "ncclCommInitRank", api->GetErrorString ? api->GetErrorString(r) : "failed");
    delete c;
    return nullptr;
  }
  return c;
}

void neptune_hip_slab_comm_destroy(neptune_hip_slab_comm_t* comm) {
  if (!comm) return;
  slab::RcclApi* api = slab::rccl();
  if (api && comm->comm) (void)api->CommDestroy(comm->comm);
  delete comm;
}

const char* neptune_hip_slab_last_error(void) { return slab::g_last_error; }

int neptune_hip_halo_exchange(neptune_hip_slab_comm_t* comm, void* field, size_t plane_bytes, int64_t n_own, int r_lo,
                              int r_hi, int peer_lo, int peer_hi, void* stream) {
  return slab::exchange(comm, field, plane_bytes, n_own, r_lo, r_hi, peer_lo, peer_hi, reinterpret_cast<hipStream_t>(stream));
}

neptune_hip_slab_plan_t* neptune_hip_slab_plan_create(neptune_hip_slab_comm_t* comm, neptune_hip_apply_fn fn, int body,
                                                      int dtype, const neptune_hip_apply_geom_t* local, int radius,
                                                      int r_lo, int r_hi, int peer_lo, int peer_hi,
                                                      const neptune_hip_launch_cfg_t* cfg) {
  auto fail = [](const char* why) -> neptune_hip_slab_plan_t* {
    slab::set_error(nullptr, "neptune_hip_slab_plan_create", why);
    return nullptr;
  };
  if (!local || geom_validate(local) != NEPTUNE_HIP_OK) return fail("malformed local geometry");
  if (!fn && (body < 0 || body >= NEPTUNE_HIP_BODY_COUNT)) return fail("neither an apply entry nor a built-in body");
  if (dtype != NEPTUNE_HIP_F64 && dtype != NEPTUNE_HIP_F32) return fail("bad element type");
  if (radius < 0 || r_lo < 0 || r_hi < 0) return fail("negative radius / ghost count");
  if ((r_lo > 0 || r_hi > 0) && !comm) return fail("ghost planes without a communicator");
  if (comm && ((r_lo > 0 && (peer_lo < 0 || peer_lo >= comm->world)) || (r_hi > 0 && (peer_hi < 0 || peer_hi >= comm->world))))
    return fail("neighbour rank outside the communicator");
  const int64_t n0 = local->out_ub[0] - local->out_lb[0];
  const int64_t n_own = n0 - r_lo - r_hi;
  if (n_own <= 0 || r_lo > n_own || r_hi > n_own) return fail("slab thinner than its halo");
  if ((r_lo > 0 && r_lo < radius) || (r_hi > 0 && r_hi < radius)) return fail("fewer ghost planes than the apply's reach");
  auto* p = new neptune_hip_slab_plan;
  p->comm = comm;
  p->fn = fn;
  p->body = body;
  p->num_inputs = local->num_inputs;
  p->esize = dtype == NEPTUNE_HIP_F64 ? 8 : 4;
  p->r_lo = r_lo; p->r_hi = r_hi; p->peer_lo = peer_lo; p->peer_hi = peer_hi;
  p->n_own = n_own;
  for (int k = 0; k < local->num_inputs; ++k) {
    if (local->in_lb[k][0] != local->out_lb[0] || local->in_ub[k][0] != local->out_ub[0]) {
      delete p;
      return fail("every input must cover the same planes as the result (owned + ghost)");
    }
    size_t b = (size_t)p->esize;
    for (int d = 1; d < local->rank; ++d) b *= (size_t)(local->in_ub[k][d] - local->in_lb[k][d]);
    p->plane_bytes[k] = b;
  }
  if (cfg) { p->cfg = *cfg; p->has_cfg = true; }
  // regions along dim 0 (result-physical planes): owned = [r_lo, r_lo + n_own); the `radius` planes next to a
  // neighbour need its ghost data, the rest is interior
  auto region = [&](int64_t a, int64_t b) {
    neptune_hip_apply_geom_t g = *local;
    for (int d = 0; d < g.rank; ++d) { g.region_lb[d] = 0; g.region_ub[d] = g.out_ub[d] - g.out_lb[d]; }
    g.region_lb[0] = a;
    g.region_ub[0] = b;
    return g;
  };
  const int64_t lo = r_lo, hi = r_lo + n_own;
  const int64_t ilo = lo + (r_lo > 0 ? radius : 0), ihi = hi - (r_hi > 0 ? radius : 0);
  p->whole = region(lo, hi);
  if (ihi <= ilo) {  // thinner than two halos: everything waits for the exchange
    p->has_interior = false;
    p->edges[p->n_edges++] = region(lo, hi);
  } else {
    p->has_interior = true;
    p->interior = region(ilo, ihi);
    if (r_lo > 0 && radius > 0) p->edges[p->n_edges++] = region(lo, ilo);
    if (r_hi > 0 && radius > 0) p->edges[p->n_edges++] = region(ihi, hi);
  }
  if (r_lo > 0 || r_hi > 0) {
    if (hipStreamCreateWithFlags(&p->comm_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&p->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p->halo_done, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      neptune_hip_slab_plan_destroy(p);
      return fail("cannot create the communication stream / events");
    }
  }
  return p;
}

void neptune_hip_slab_plan_destroy(neptune_hip_slab_plan_t* p) {
  if (!p) return;
  if (p->comm_stream) { (void)hipStreamSynchronize(p->comm_stream); (void)hipStreamDestroy(p->comm_stream); }
  if (p->ready) (void)hipEventDestroy(p->ready);
  if (p->halo_done) (void)hipEventDestroy(p->halo_done);
  delete p;
}

int neptune_hip_slab_apply(neptune_hip_slab_plan_t* p, const void* const* in, void* out, void* compute_stream, int overlap) {
  if (!p || !in || !out) return NEPTUNE_HIP_EINVAL;
  for (int k = 0; k < p->num_inputs; ++k)
    if (!in[k]) return NEPTUNE_HIP_EINVAL;
  hipStream_t cs = reinterpret_cast<hipStream_t>(compute_stream);
  const neptune_hip_launch_cfg_t* cfg = p->has_cfg ? &p->cfg : nullptr;
  auto launch = [&](const neptune_hip_apply_geom_t& g) {
    return p->fn ? p->fn(&g, in, out, compute_stream, cfg) : neptune_hip_apply_builtin(p->body, &g, in, out, compute_stream, cfg);
  };
  if (p->r_lo == 0 && p->r_hi == 0) return launch(p->whole);
  // 1. the exchange, on the communication stream, once the input is complete on the compute stream
  NEPTUNE_HIP_TRY(p->comm, hipEventRecord(p->ready, cs));
  NEPTUNE_HIP_TRY(p->comm, hipStreamWaitEvent(p->comm_stream, p->ready, 0));
  for (int k = 0; k < p->num_inputs; ++k) {
    const int rc = slab::exchange(p->comm, const_cast<void*>(in[k]), p->plane_bytes[k], p->n_own, p->r_lo, p->r_hi, p->peer_lo,
                                  p->peer_hi, p->comm_stream);
    if (rc != NEPTUNE_HIP_OK) return rc;
  }
  NEPTUNE_HIP_TRY(p->comm, hipEventRecord(p->halo_done, p->comm_stream));
  if (!overlap) NEPTUNE_HIP_TRY(p->comm, hipStreamWaitEvent(cs, p->halo_done, 0));
  // 2. interior planes overlap the exchange
  if (p->has_interior) {
    const int rc = launch(p->interior);
    if (rc != NEPTUNE_HIP_OK) return rc;
  }
  // 3. edge planes once the ghosts have landed
  NEPTUNE_HIP_TRY(p->comm, hipStreamWaitEvent(cs, p->halo_done, 0));
  for (int e = 0; e < p->n_edges; ++e) {
    const int rc = launch(p->edges[e]);
    if (rc != NEPTUNE_HIP_OK) return rc;
  }
  return NEPTUNE_HIP_OK;
}


}  // extern "C"

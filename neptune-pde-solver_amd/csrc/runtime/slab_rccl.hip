// slab_rccl.hip -- include/neptune_hip.h section 8: halo exchange (RCCL or peer copies) and the sharded apply plan.  Host
// code plus the one-wave handshake kernel of the peer transport; its own translation unit so that it builds in seconds,
// linked into libneptune_hip.so.
#include "slab_rccl.hpp"

#include "../kernels/apply_launch.hpp"   // geom_validate (no kernel is instantiated here)

using namespace neptune_hip;

// ---------------------------------------------------------------- peer transport: the handshake kernel
// One wave.  Lane 0 stores the counters its neighbours wait for (release, system scope: after every copy the stream
// ran before this kernel), then polls its own mailbox until the neighbours' counters arrive (acquire, system scope).
// The mailbox is host memory every process registered (slab_peer.hpp), so the words are coherent across devices and
// processes by construction.  Bounded: after `timeout_ticks` of the constant 100 MHz clock the wave sets the error
// word and retires -- a rank whose neighbour died drains its stream instead of hanging the GPU.
__global__ void neptune_peer_signal_wait(uint64_t* sig_lo, uint64_t* sig_hi, const uint64_t* wait_lo, const uint64_t* wait_hi,
                                         uint64_t v_lo, uint64_t v_hi, uint64_t timeout_ticks, uint32_t* err, uint32_t code) {
  if (threadIdx.x != 0) return;
  if (sig_lo) __hip_atomic_store(sig_lo, v_lo, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  if (sig_hi) __hip_atomic_store(sig_hi, v_hi, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  const uint64_t t0 = wall_clock64();
  const uint64_t* w[2] = {wait_lo, wait_hi};
  const uint64_t v[2] = {v_lo, v_hi};
  for (int s = 0; s < 2; ++s) {
    if (!w[s]) continue;
    while (__hip_atomic_load(w[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v[s]) {
      if (wall_clock64() - t0 > timeout_ticks) {
        __hip_atomic_fetch_or(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
      }
      __builtin_amdgcn_s_sleep(16);
    }
  }
  __threadfence_system();
}

namespace neptune_hip {
namespace slab {
void peer_signal_wait(hipStream_t stream, uint64_t* sig_lo, uint64_t* sig_hi, const uint64_t* wait_lo, const uint64_t* wait_hi,
                      uint64_t v_lo, uint64_t v_hi, uint32_t* err, uint32_t code) {
  const uint64_t ticks = (uint64_t)(peer::timeout_s() * 1e8);   // wall_clock64 counts at 100 MHz on gfx950
  hipLaunchKernelGGL(neptune_peer_signal_wait, dim3(1), dim3(64), 0, stream, sig_lo, sig_hi, wait_lo, wait_hi, v_lo, v_hi, ticks, err, code);
  (void)hipGetLastError();
}
}  // namespace slab
}  // namespace neptune_hip

extern "C" {

// ---------------------------------------------------------------- slab decomposition over RCCL
struct neptune_hip_slab_comm : neptune_hip::slab::Comm {};
struct neptune_hip_slab_plan : neptune_hip::slab::Plan {};

int neptune_hip_slab_unique_id_ex(int transport, void* id_out) {
  if (!id_out) return NEPTUNE_HIP_EINVAL;
  if (transport == NEPTUNE_HIP_TRANSPORT_PEER) {
    // 128 random bytes: the first 16 name the node-local shared-memory segment of the communicator
    unsigned char* b = static_cast<unsigned char*>(id_out);
    FILE* f = fopen("/dev/urandom", "rb");
    const size_t got = f ? fread(b, 1, NEPTUNE_HIP_SLAB_ID_BYTES, f) : 0;
    if (f) fclose(f);
    if (got != NEPTUNE_HIP_SLAB_ID_BYTES) {
      uint64_t x = (uint64_t)getpid() * 0x9E3779B97F4A7C15ull ^ (uint64_t)(slab::peer::now_s() * 1e9);
      for (int i = 0; i < NEPTUNE_HIP_SLAB_ID_BYTES; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; b[i] = (unsigned char)(x >> 24); }
    }
    return NEPTUNE_HIP_OK;
  }
  if (transport != NEPTUNE_HIP_TRANSPORT_RCCL) return NEPTUNE_HIP_EINVAL;
  slab::RcclApi* api = slab::rccl();
  if (!api) { slab::set_error(nullptr, "rccl", "librccl.so.1 is not loadable"); return NEPTUNE_HIP_ECOMM; }
  ncclUniqueId id;
  static_assert(sizeof(id) == NEPTUNE_HIP_SLAB_ID_BYTES, "ncclUniqueId size");
  NEPTUNE_RCCL_TRY(nullptr, api, api->GetUniqueId(&id));
  memcpy(id_out, &id, sizeof id);
  return NEPTUNE_HIP_OK;
}
int neptune_hip_slab_unique_id(void* id_out) { return neptune_hip_slab_unique_id_ex(NEPTUNE_HIP_TRANSPORT_RCCL, id_out); }

neptune_hip_slab_comm_t* neptune_hip_slab_comm_create_ex(int transport, const void* id, int rank, int world) {
  if (world < 1 || rank < 0 || rank >= world || (!id && world > 1) ||
      (transport != NEPTUNE_HIP_TRANSPORT_RCCL && transport != NEPTUNE_HIP_TRANSPORT_PEER)) {
    slab::set_error(nullptr, "neptune_hip_slab_comm_create", "bad transport / rank / world / id");
    return nullptr;
  }
  if (transport == NEPTUNE_HIP_TRANSPORT_PEER) {
    unsigned char own_id[NEPTUNE_HIP_SLAB_ID_BYTES];
    if (!id) { (void)neptune_hip_slab_unique_id_ex(transport, own_id); id = own_id; }
    char why[256] = "";
    slab::peer::State* st = slab::peer::create(id, rank, world, why, sizeof why);
    if (!st) { slab::set_error(nullptr, "peer transport", why); return nullptr; }
    auto* c = new neptune_hip_slab_comm;
    c->transport = transport;
    c->rank = rank;
    c->world = world;
    c->peer = st;
    return c;
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
neptune_hip_slab_comm_t* neptune_hip_slab_comm_create(const void* id, int rank, int world) {
  return neptune_hip_slab_comm_create_ex(NEPTUNE_HIP_TRANSPORT_RCCL, id, rank, world);
}

void neptune_hip_slab_comm_destroy(neptune_hip_slab_comm_t* comm) {
  if (!comm) return;
  if (comm->peer) {
    (void)hipDeviceSynchronize();   // no handshake kernel or copy of this communicator may still be in flight
    slab::peer::destroy(comm->peer);
  } else {
    slab::RcclApi* api = slab::rccl();
    if (api && comm->comm) (void)api->CommDestroy(comm->comm);
  }
  delete comm;
}

const char* neptune_hip_slab_comm_transport(const neptune_hip_slab_comm_t* comm) {
  if (!comm) return "none";
  return comm->transport == NEPTUNE_HIP_TRANSPORT_PEER ? "peer" : "rccl";
}

int neptune_hip_slab_comm_status(neptune_hip_slab_comm_t* comm) {
  if (!comm) return NEPTUNE_HIP_EINVAL;
  if (comm->peer && *comm->peer->err_host) {
    slab::set_error(comm, "peer transport", (*comm->peer->err_host & 1) ? "a device-side wait timed out: a neighbour never freed its ghost planes"
                                                                        : "a device-side wait timed out: a neighbour's planes never arrived");
    return NEPTUNE_HIP_ECOMM;
  }
  return NEPTUNE_HIP_OK;
}

const char* neptune_hip_slab_last_error(void) { return slab::g_last_error; }

int neptune_hip_halo_exchange(neptune_hip_slab_comm_t* comm, void* field, size_t plane_bytes, int64_t n_own, int r_lo,
                              int r_hi, int peer_lo, int peer_hi, void* stream) {
  return slab::exchange(comm, field, plane_bytes, n_own, r_lo, r_hi, peer_lo, peer_hi, reinterpret_cast<hipStream_t>(stream));
}
int neptune_hip_halo_exchange_many(neptune_hip_slab_comm_t* comm, void* const* fields, const size_t* plane_bytes, int nfields,
                                   int64_t n_own, int r_lo, int r_hi, int peer_lo, int peer_hi, void* stream) {
  if (nfields < 1 || nfields > NEPTUNE_HIP_MAX_INPUTS || !plane_bytes) return NEPTUNE_HIP_EINVAL;
  return slab::exchange_many(comm, fields, plane_bytes, nfields, n_own, r_lo, r_hi, peer_lo, peer_hi, reinterpret_cast<hipStream_t>(stream));
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
    // Priority of the communication stream.  The exchange and the interior launch become runnable at the same instant
    // (both wait for the previous step) and the interior grid fills every CU, so the exchange's kernels must not be
    // dispatched BEHIND it, or the exchange runs after the interior and the edge launches wait for it.  Peer transport:
    // greatest priority -- its two handshake waves cost nothing and the copies run on the SDMA engines.  RCCL: default
    // priority -- its copy kernels at the greatest priority took the interior's CUs on one GPU (loop-back, 1024^3 slab
    // of 8: step 0.56 ms against 0.39 ms; both ran beside the interior, profiles/r03_multigpu.txt); between devices the
    // exchange is bound by the link and may want the head start: NEPTUNE_HIP_COMM_PRIORITY=high|normal overrides, and
    // bench.py times both.
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
    const char* pe = getenv("NEPTUNE_HIP_COMM_PRIORITY");
    const bool high = pe && *pe ? strcmp(pe, "normal") != 0 : (comm && comm->transport == NEPTUNE_HIP_TRANSPORT_PEER);
    if (!high) greatest = 0;
    if (hipStreamCreateWithPriority(&p->comm_stream, hipStreamNonBlocking, greatest) != hipSuccess ||
        hipEventCreateWithFlags(&p->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p->halo_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p->edges_done, hipEventDisableTiming) != hipSuccess) {
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
  if (p->edges_done) (void)hipEventDestroy(p->edges_done);
  if (p->tev) {
    for (int i = 0; i < slab::Plan::kTimingRing; ++i)
      for (hipEvent_t e : {p->tev[i].x0, p->tev[i].x1, p->tev[i].i0, p->tev[i].i1, p->tev[i].e1})
        if (e) (void)hipEventDestroy(e);
    delete[] p->tev;
  }
  delete p;
}

int neptune_hip_slab_plan_timing(neptune_hip_slab_plan_t* p, int on) {
  if (!p) return NEPTUNE_HIP_EINVAL;
  if (on && !p->tev && (p->r_lo > 0 || p->r_hi > 0)) {
    p->tev = new slab::Plan::StepEvents[slab::Plan::kTimingRing]();
    for (int i = 0; i < slab::Plan::kTimingRing; ++i)
      for (hipEvent_t* e : {&p->tev[i].x0, &p->tev[i].x1, &p->tev[i].i0, &p->tev[i].i1, &p->tev[i].e1})
        NEPTUNE_HIP_TRY(p->comm, hipEventCreate(e));
  }
  p->timing = on != 0 && p->tev != nullptr;
  p->timed_steps = 0;
  return NEPTUNE_HIP_OK;
}

int neptune_hip_slab_plan_timing_read(neptune_hip_slab_plan_t* p, double out[6]) {
  if (!p || !out) return NEPTUNE_HIP_EINVAL;
  for (int i = 0; i < 6; ++i) out[i] = 0.0;
  if (!p->tev || p->timed_steps <= 0) return NEPTUNE_HIP_OK;
  const int64_t n = p->timed_steps < slab::Plan::kTimingRing ? p->timed_steps : slab::Plan::kTimingRing;
  double acc[5] = {0, 0, 0, 0, 0};
  for (int64_t k = 0; k < n; ++k) {
    const slab::Plan::StepEvents& ev = p->tev[(p->timed_steps - 1 - k) % slab::Plan::kTimingRing];
    NEPTUNE_HIP_TRY(p->comm, hipEventSynchronize(ev.e1));
    NEPTUNE_HIP_TRY(p->comm, hipEventSynchronize(ev.x1));
    float exch = 0, inter = 0, i1_to_x1 = 0, step = 0, x1_to_e1 = 0, i1_to_e1 = 0;
    NEPTUNE_HIP_TRY(p->comm, hipEventElapsedTime(&exch, ev.x0, ev.x1));
    NEPTUNE_HIP_TRY(p->comm, hipEventElapsedTime(&inter, ev.i0, ev.i1));
    NEPTUNE_HIP_TRY(p->comm, hipEventElapsedTime(&step, ev.i0, ev.e1));
    NEPTUNE_HIP_TRY(p->comm, hipEventElapsedTime(&i1_to_e1, ev.i1, ev.e1));
    NEPTUNE_HIP_TRY(p->comm, hipEventElapsedTime(&x1_to_e1, ev.x1, ev.e1));
    i1_to_x1 = i1_to_e1 - x1_to_e1;                    // > 0: the exchange ended after the interior -- the edges waited
    acc[0] += exch;
    acc[1] += inter;
    acc[2] += i1_to_x1 > 0 ? i1_to_x1 : 0;
    acc[3] += i1_to_x1 > 0 ? x1_to_e1 : i1_to_e1;      // the edge launches themselves
    acc[4] += step;
  }
  for (int i = 0; i < 5; ++i) out[i] = acc[i] / (double)n;
  out[5] = (double)n;
  return NEPTUNE_HIP_OK;
}

int neptune_hip_slab_apply(neptune_hip_slab_plan_t* p, const void* const* in, void* out, void* compute_stream, int overlap) {
  if (!p || !in || !out) return NEPTUNE_HIP_EINVAL;
  for (int k = 0; k < p->num_inputs; ++k)
    if (!in[k]) return NEPTUNE_HIP_EINVAL;
  hipStream_t cs = reinterpret_cast<hipStream_t>(compute_stream);
  const neptune_hip_launch_cfg_t* cfg = p->has_cfg ? &p->cfg : nullptr;
  auto launch_on = [&](const neptune_hip_apply_geom_t& g, void* st) {
    return p->fn ? p->fn(&g, in, out, st, cfg) : neptune_hip_apply_builtin(p->body, &g, in, out, st, cfg);
  };
  auto launch = [&](const neptune_hip_apply_geom_t& g) { return launch_on(g, compute_stream); };
  if (p->r_lo == 0 && p->r_hi == 0) return launch(p->whole);
  const slab::Plan::StepEvents* ev = p->timing ? &p->tev[p->timed_steps % slab::Plan::kTimingRing] : nullptr;
  // 1. the exchange, on the communication stream, once the input is complete on the compute stream
  NEPTUNE_HIP_TRY(p->comm, hipEventRecord(p->ready, cs));
  NEPTUNE_HIP_TRY(p->comm, hipStreamWaitEvent(p->comm_stream, p->ready, 0));
  if (ev) NEPTUNE_HIP_TRY(p->comm, hipEventRecord(ev->x0, p->comm_stream));
  {
    void* fields[NEPTUNE_HIP_MAX_INPUTS];
    for (int k = 0; k < p->num_inputs; ++k) fields[k] = const_cast<void*>(in[k]);
    const int rc = slab::exchange_many(p->comm, fields, p->plane_bytes, p->num_inputs, p->n_own, p->r_lo, p->r_hi, p->peer_lo,
                                       p->peer_hi, p->comm_stream);
    if (rc != NEPTUNE_HIP_OK) return rc;
  }
  // (with the edge launches on the communication stream nobody waits for the exchange alone: one event less per step)
  const bool edges_on_comm = overlap && p->n_edges > 0 && !slab::edges_on_compute_stream();
  if (!edges_on_comm) NEPTUNE_HIP_TRY(p->comm, hipEventRecord(p->halo_done, p->comm_stream));
  if (ev) NEPTUNE_HIP_TRY(p->comm, hipEventRecord(ev->x1, p->comm_stream));
  if (!overlap) NEPTUNE_HIP_TRY(p->comm, hipStreamWaitEvent(cs, p->halo_done, 0));
  // 2. interior planes overlap the exchange
  if (ev) NEPTUNE_HIP_TRY(p->comm, hipEventRecord(ev->i0, cs));
  if (p->has_interior) {
    const int rc = launch(p->interior);
    if (rc != NEPTUNE_HIP_OK) return rc;
  }
  if (ev) NEPTUNE_HIP_TRY(p->comm, hipEventRecord(ev->i1, cs));
  // 3. edge planes once the ghosts have landed.  With overlap they are launched on the COMMUNICATION stream, right behind the
  //    exchange: they read the input and write planes of their own, so they need not wait for the interior launch -- their
  //    workgroups take the CUs the interior's last workgroups leave (a launch ends over ~5 % of its duration,
  //    profiles/r03_timeline.txt) instead of starting one after the other behind it.  The compute stream joins at the end.
  if (edges_on_comm) {
    for (int e = 0; e < p->n_edges; ++e) {
      const int rc = launch_on(p->edges[e], p->comm_stream);
      if (rc != NEPTUNE_HIP_OK) return rc;
    }
    NEPTUNE_HIP_TRY(p->comm, hipEventRecord(p->edges_done, p->comm_stream));
    NEPTUNE_HIP_TRY(p->comm, hipStreamWaitEvent(cs, p->edges_done, 0));
  } else {
    NEPTUNE_HIP_TRY(p->comm, hipStreamWaitEvent(cs, p->halo_done, 0));
    for (int e = 0; e < p->n_edges; ++e) {
      const int rc = launch(p->edges[e]);
      if (rc != NEPTUNE_HIP_OK) return rc;
    }
  }
  if (ev) {
    NEPTUNE_HIP_TRY(p->comm, hipEventRecord(ev->e1, cs));
    ++p->timed_steps;
  }
  return NEPTUNE_HIP_OK;
}


}  // extern "C"

// neptune_hip_rt.hip -- implementation of include/neptune_hip.h: the thin C-ABI runtime the
// emitted host code, the Python frontend and the bench call into.  gfx950 only.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared
//   -ffp-contract=off is part of the contract: the reference evaluates body ops one by one in
//   strict IEEE with no FMA (lib/Pipeline/NeptuneIRPassesPipeline.cpp:28-46 has no fusing
//   pass; lib/Compiler/NeptuneCompiler.cpp:332-337 targets cpu "generic"), and parity with
//   it is bit-exact only if the device code does the same.
#define NEPTUNE_HIP_FULL_VARIANTS 1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "../../../include/neptune_hip.h"
#include "../kernels/apply_launch.hpp"
#include "../kernels/util_kernels.hpp"
#include "rt_bodies.hpp"

using namespace neptune_hip;

namespace {

struct RuntimeState {
  std::mutex mu;
  bool inited = false;
  int device = -1;
  char arch[256] = "unknown";
  int cus = 0;
  unsigned long long* counter = nullptr;  // device word for count_mismatch
  void* reduce_ws = nullptr;              // kReduceBlocks partials + 1 result (8 B each)
  // idle device blocks kept for the temporaries of lowered functions (neptune_hip_pool_*)
  struct PoolBlock { void* p; size_t bytes; };
  std::vector<PoolBlock> pool;
  size_t pool_bytes = 0;
  long long pool_cap = -1;                // bytes the pool may hold; -1 = not yet decided
};
RuntimeState& rt() {
  static RuntimeState s;
  return s;
}

[[noreturn]] void die(const char* what) {
  fprintf(stderr, "[NeptuneRT][HIP] %s\n", what);
  abort();
}

void ensure_init() {
  RuntimeState& s = rt();
  if (s.inited) return;
  neptune_hip_init(s.device < 0 ? 0 : s.device);
}

inline hipStream_t as_stream(void* p) { return reinterpret_cast<hipStream_t>(p); }

// the built-in bodies live in their own translation units (rt_body_*.hip)
const rtbody::Entry* body_entry(int body) {
  switch (body) {
    case NEPTUNE_HIP_BODY_LAP2D5_F64: return &rtbody::lap2d5();
    case NEPTUNE_HIP_BODY_LAP3D7_F64: return &rtbody::lap3d7();
    case NEPTUNE_HIP_BODY_LAP3D27_F32: return &rtbody::lap3d27();
    case NEPTUNE_HIP_BODY_LAP1D3_F64: return &rtbody::lap1d3();
  }
  return nullptr;
}

// out must not overlap an input: every wave reads neighbours other waves may already have
// overwritten.  (The reference materialises a fresh buffer per apply, DataflowLowering.cpp:281.)
bool overlaps(const void* a, size_t na, const void* b, size_t nb) {
  const uintptr_t x = (uintptr_t)a, y = (uintptr_t)b;
  return x < y + nb && y < x + na;
}
size_t geom_bytes(const int64_t* lb, const int64_t* ub, int rank, size_t elem) {
  size_t n = elem;
  for (int d = 0; d < rank; ++d) n *= (size_t)(ub[d] - lb[d]);
  return n;
}
int check_no_alias(const neptune_hip_apply_geom_t* g, const void* const* in, const void* out, size_t elem) {
  const size_t ob = geom_bytes(g->out_lb, g->out_ub, g->rank, elem);
  for (int k = 0; k < g->num_inputs; ++k) {
    const size_t ib = geom_bytes(g->in_lb[k], g->in_ub[k], g->rank, elem);
    if (overlaps(in[k], ib, out, ob)) return NEPTUNE_HIP_EINVAL;
  }
  return NEPTUNE_HIP_OK;
}

size_t body_elem_size(int body) { return body == NEPTUNE_HIP_BODY_LAP3D27_F32 ? 4 : 8; }

}  // namespace

namespace {
// average milliseconds per launch of `launch(cfg)` on `st` (HIP events; blocking); negative = the launch's error code
template <class L>
double time_launches(L&& launch, hipStream_t st, const neptune_hip_launch_cfg_t* cfg, int warmup, int reps) {
  if (reps <= 0) return -1.0;
  for (int i = 0; i < warmup; ++i) {
    const int rc = launch(cfg);
    if (rc != NEPTUNE_HIP_OK) return (double)rc;
  }
  hipEvent_t e0, e1;
  NEPTUNE_HIP_CHECK(hipEventCreate(&e0));
  NEPTUNE_HIP_CHECK(hipEventCreate(&e1));
  NEPTUNE_HIP_CHECK(hipEventRecord(e0, st));
  int rc = NEPTUNE_HIP_OK;
  for (int i = 0; i < reps && rc == NEPTUNE_HIP_OK; ++i) rc = launch(cfg);
  NEPTUNE_HIP_CHECK(hipEventRecord(e1, st));
  NEPTUNE_HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  NEPTUNE_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  NEPTUNE_HIP_CHECK(hipEventDestroy(e0));
  NEPTUNE_HIP_CHECK(hipEventDestroy(e1));
  return rc != NEPTUNE_HIP_OK ? (double)rc : (double)ms / reps;
}

// Plan-time tuning shared by the built-in bodies and lowered applies: the automatic launch first, then every
// march tile `nv` the code object behind `launch` holds x a few chunk lengths.  All candidates compute the same
// bits, so the timed launches leave `out` exactly as a normal launch would.
// Two passes: every candidate once (reps launches), then a play-off of the four fastest, measured again in turn with
// three times the repetitions -- the first pass runs while the clocks are still settling (candidates measured late look
// faster than they are: a 27-point workload once picked the last tile of the list at 0.189 ms and then ran at 0.199),
// the play-off compares like with like.
template <class L>
int autotune_launches(L&& launch, bool march_planned, int rank, int nv, hipStream_t st, int reps,
                      neptune_hip_launch_cfg_t* best, double* best_ms) {
  if (reps <= 0) reps = 5;
  neptune_hip_launch_cfg_t probe = {NEPTUNE_HIP_KERNEL_AUTO, -1, 0, 0};
  *best = probe;
  double best_t = time_launches(launch, st, &probe, 2, reps);
  if (best_t < 0) return (int)best_t;
  // short launches need more repetitions to tell tiles 2-3 % apart: at least ~4 ms of launches per candidate
  if (best_t > 0 && best_t * reps < 4.0) reps = (int)(4.0 / best_t) + 1 < 24 ? (int)(4.0 / best_t) + 1 : 24;
  struct Cand { neptune_hip_launch_cfg_t cfg; double t; };
  std::vector<Cand> cands;
  cands.push_back({probe, best_t});
  if (march_planned) {
    const int chunks3[] = {0, 32, 64, 128, 256, 512}, chunks2[] = {0};
    for (int v = 0; v < nv; ++v) {
      const MarchVariant* mv = march_variant(rank, v);   // indices below the caller's count name the same tiles everywhere
      if (!mv) break;
      const bool tile2 = rank == 2 && mv->jk;
      const int* chunks = (rank == 3 || !tile2) ? chunks3 : chunks2;
      const int nc = (rank == 3 || !tile2) ? 6 : 1;
      for (int c = 0; c < nc; ++c) {
        neptune_hip_launch_cfg_t cfg = {NEPTUNE_HIP_KERNEL_MARCH, v, chunks[c], 0};
        const double t = time_launches(launch, st, &cfg, 1, reps);
        if (t > 0) cands.push_back({cfg, t});
      }
    }
    std::sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) { return a.t < b.t; });
    const size_t top = cands.size() < 4 ? cands.size() : 4;
    for (size_t i = 0; i < top; ++i) cands[i].t = 1e30;
    for (int round = 0; round < 2; ++round)
      for (size_t i = 0; i < top; ++i) {
        const double t = time_launches(launch, st, &cands[i].cfg, 1, 3 * reps);
        if (t > 0 && t < cands[i].t) cands[i].t = t;
      }
    size_t win = 0;
    for (size_t i = 1; i < top; ++i)
      if (cands[i].t < cands[win].t) win = i;
    *best = cands[win].cfg;
    best_t = cands[win].t;
  }
  if (best_ms) *best_ms = best_t;
  return NEPTUNE_HIP_OK;
}
}  // namespace


namespace {
template <bool XPAY>
int vec_update(int dtype, int64_t n, double a, const void* x, void* y, void* stream) {
  if (!x || !y || n < 0) return NEPTUNE_HIP_EINVAL;
  if (n == 0) return NEPTUNE_HIP_OK;
  if (dtype != NEPTUNE_HIP_F64 && dtype != NEPTUNE_HIP_F32) return NEPTUNE_HIP_EINVAL;
  // the kernel reads x and writes y through __restrict__ pointers: overlapping ranges would be undefined behaviour
  if (overlaps(x, (size_t)n * (dtype == NEPTUNE_HIP_F64 ? 8 : 4), y, (size_t)n * (dtype == NEPTUNE_HIP_F64 ? 8 : 4))) return NEPTUNE_HIP_EINVAL;
  ensure_init();
  if (((uintptr_t)x | (uintptr_t)y) % 16 == 0) {   // 16-byte vectors, exact grid
    const int vk = dtype == NEPTUNE_HIP_F64 ? 2 : 4;
    const int64_t nv = n / vk, vblocks = nv > 0 ? (nv + 255) / 256 : 1;
    if (vblocks <= 0x7fffffffLL) {
      if (dtype == NEPTUNE_HIP_F64)
        hipLaunchKernelGGL((neptune_vec_update_v<double, XPAY>), dim3((uint32_t)vblocks), dim3(256), 0, as_stream(stream), n, a, (const double*)x, (double*)y);
      else
        hipLaunchKernelGGL((neptune_vec_update_v<float, XPAY>), dim3((uint32_t)vblocks), dim3(256), 0, as_stream(stream), n, (float)a, (const float*)x, (float*)y);
      NEPTUNE_HIP_CHECK(hipGetLastError());
      return NEPTUNE_HIP_OK;
    }
  }
  const int64_t want = (n + 255) / 256;
  const uint32_t blocks = (uint32_t)(want < 256 * 32 ? want : 256 * 32);
  if (dtype == NEPTUNE_HIP_F64)
    hipLaunchKernelGGL((neptune_vec_update<double, XPAY>), dim3(blocks), dim3(256), 0, as_stream(stream), n, a, (const double*)x, (double*)y);
  else if (dtype == NEPTUNE_HIP_F32)
    hipLaunchKernelGGL((neptune_vec_update<float, XPAY>), dim3(blocks), dim3(256), 0, as_stream(stream), n, (float)a, (const float*)x, (float*)y);
  else
    return NEPTUNE_HIP_EINVAL;
  NEPTUNE_HIP_CHECK(hipGetLastError());
  return NEPTUNE_HIP_OK;
}
}  // namespace

extern "C" {

// ---------------------------------------------------------------- runtime
void neptune_hip_init(int device) {
  RuntimeState& s = rt();
  std::lock_guard<std::mutex> lk(s.mu);
  if (s.inited && s.device == device) return;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) die("no HIP device available");
  if (device < 0 || device >= count) die("neptune_hip_init: device index out of range");
  NEPTUNE_HIP_CHECK(hipSetDevice(device));
  hipDeviceProp_t prop;
  NEPTUNE_HIP_CHECK(hipGetDeviceProperties(&prop, device));
  strncpy(s.arch, prop.gcnArchName, sizeof(s.arch) - 1);
  s.cus = prop.multiProcessorCount;
  if (strncmp(s.arch, "gfx950", 6) != 0)
    fprintf(stderr, "[NeptuneRT][HIP] warning: built for gfx950, device reports %s\n", s.arch);
  if (!s.counter) NEPTUNE_HIP_CHECK(hipMalloc((void**)&s.counter, sizeof(unsigned long long)));
  if (!s.reduce_ws) NEPTUNE_HIP_CHECK(hipMalloc(&s.reduce_ws, (kReduceBlocks + 1) * 8));
  s.device = device;
  s.inited = true;
}

// state of neptune_hip_step_loop (defined further down)
namespace {
struct LoopKey {
  int applies, from0;         // (graph cache only) what one node of the cached graph is, and which field the loop started in
  neptune_hip_apply_fn fn3;   // three chained applies in one launch (<tag>__geom3), or nullptr
  neptune_hip_apply_fn fn2;   // two chained applies in one launch (a lowered apply's <tag>__geom2), or nullptr
  neptune_hip_apply_fn fn;
  int body;
  neptune_hip_apply_geom_t g;
  void* fields[2];
  const void* in[NEPTUNE_HIP_MAX_INPUTS];
  neptune_hip_launch_cfg_t cfg;
  hipStream_t stream;
};
struct LoopGraph {
  LoopKey key;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  uint64_t stamp = 0;
};
constexpr int kLoopGraphs = 8;
LoopGraph g_loops[kLoopGraphs];
uint64_t g_loop_clock = 0;

hipStream_t g_loop_stream = nullptr;   // stands in for the legacy default stream, which cannot be captured
hipEvent_t g_loop_ev[2] = {nullptr, nullptr};
}  // namespace

void neptune_hip_finalize(void) {
  RuntimeState& s = rt();
  std::lock_guard<std::mutex> lk(s.mu);
  if (!s.inited) return;
  if (s.counter) {
    (void)hipFree(s.counter);
    s.counter = nullptr;
  }
  if (s.reduce_ws) {
    (void)hipFree(s.reduce_ws);
    s.reduce_ws = nullptr;
  }
  for (auto& e : g_loops)
    if (e.exec) {
      (void)hipGraphExecDestroy(e.exec);
      (void)hipGraphDestroy(e.graph);
      e.exec = nullptr;
      e.graph = nullptr;
      e.stamp = 0;
    }
  for (auto& b : s.pool) (void)hipFree(b.p);
  s.pool.clear();
  s.pool_bytes = 0;
  s.inited = false;
}

int neptune_hip_available(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) return 0;
  return count > 0 ? 1 : 0;
}

const char* neptune_hip_arch(void) {
  ensure_init();
  return rt().arch;
}
int neptune_hip_cu_count(void) {
  ensure_init();
  return rt().cus;
}
const char* neptune_hip_version(void) { return "neptune-hip 0.1 (gfx950, ffp-contract=off)"; }

void* neptune_hip_malloc(size_t bytes) {
  ensure_init();
  void* p = nullptr;
  if (bytes == 0) bytes = 16;
  NEPTUNE_HIP_CHECK(hipMalloc(&p, bytes));
  return p;
}
void neptune_hip_free(void* dptr) {
  if (dptr) NEPTUNE_HIP_CHECK(hipFree(dptr));
}
void neptune_hip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
  NEPTUNE_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
  NEPTUNE_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
}
void neptune_hip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
  NEPTUNE_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
  NEPTUNE_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
}
void neptune_hip_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
  NEPTUNE_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
}
void neptune_hip_stream_sync(void* stream) { NEPTUNE_HIP_CHECK(hipStreamSynchronize(as_stream(stream))); }
void neptune_hip_device_sync(void) { NEPTUNE_HIP_CHECK(hipDeviceSynchronize()); }

int neptune_hip_is_device_ptr(const void* p) {
  if (!p) return 0;
  hipPointerAttribute_t attr;
  memset(&attr, 0, sizeof(attr));
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();  // plain host memory is reported as an error; clear it
    return 0;
  }
  return attr.type == hipMemoryTypeDevice ? 1 : 0;
}

// ---------------------------------------------------------------- block pool
// hipMalloc/hipFree of a field-sized block costs far more than the kernels that use it (an 8 GiB temp:
// hundreds of milliseconds for mapping and unmapping), and a lowered function needs one for every
// apply result it cannot write straight into a destination field.  Idle blocks are therefore kept and
// handed out again: best fit within 25 % of the request, LIFO among equals.  Blocks are only returned
// by scopes that have synchronised their stream, so a cached block is never in use.
static long long pool_cap_bytes(RuntimeState& s) {
  if (s.pool_cap < 0) {
    const char* e = getenv("NEPTUNE_HIP_POOL_BYTES");
    if (e && *e) {
      s.pool_cap = atoll(e);
    } else {
      size_t fr = 0, tot = 0;
      s.pool_cap = (hipMemGetInfo(&fr, &tot) == hipSuccess) ? (long long)(tot / 4) : (long long)(16ll << 30);
    }
    if (s.pool_cap < 0) s.pool_cap = 0;
  }
  return s.pool_cap;
}

void* neptune_hip_pool_alloc(size_t bytes) {
  ensure_init();
  RuntimeState& s = rt();
  if (bytes == 0) bytes = 16;
  {
    std::lock_guard<std::mutex> lk(s.mu);
    int best = -1;
    for (int i = (int)s.pool.size() - 1; i >= 0; --i) {
      const size_t b = s.pool[i].bytes;
      if (b >= bytes && b - bytes <= bytes / 4 && (best < 0 || b < s.pool[best].bytes)) best = i;
    }
    if (best >= 0) {
      void* p = s.pool[best].p;
      s.pool_bytes -= s.pool[best].bytes;
      s.pool.erase(s.pool.begin() + best);
      return p;
    }
  }
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) {
    (void)hipGetLastError();
    neptune_hip_pool_trim();  // out of memory with idle blocks held back: give them up and retry once
    NEPTUNE_HIP_CHECK(hipMalloc(&p, bytes));
  }
  return p;
}

void neptune_hip_pool_release(void* p, size_t bytes) {
  if (!p) return;
  RuntimeState& s = rt();
  if (bytes == 0) bytes = 16;
  {
    std::lock_guard<std::mutex> lk(s.mu);
    if (s.inited && (long long)(s.pool_bytes + bytes) <= pool_cap_bytes(s)) {
      s.pool.push_back({p, bytes});
      s.pool_bytes += bytes;
      return;
    }
  }
  NEPTUNE_HIP_CHECK(hipFree(p));
}

void neptune_hip_pool_trim(void) {
  RuntimeState& s = rt();
  std::vector<RuntimeState::PoolBlock> blocks;
  {
    std::lock_guard<std::mutex> lk(s.mu);
    blocks.swap(s.pool);
    s.pool_bytes = 0;
  }
  for (auto& b : blocks) NEPTUNE_HIP_CHECK(hipFree(b.p));
}

size_t neptune_hip_pool_cached_bytes(void) {
  RuntimeState& s = rt();
  std::lock_guard<std::mutex> lk(s.mu);
  return s.pool_bytes;
}

static int64_t g_slab[4] = {0, 0, 0, 0};
static bool g_slab_set = false;
static void* g_slab_pending = nullptr;  // hipEvent_t of a halo exchange still in flight on another stream
int neptune_hip_set_slab(int64_t start, int64_t stop, int64_t ghost_lo, int64_t ghost_hi) {
  if (stop < start || ghost_lo < 0 || ghost_hi < 0) return NEPTUNE_HIP_EINVAL;
  g_slab[0] = start; g_slab[1] = stop; g_slab[2] = ghost_lo; g_slab[3] = ghost_hi;
  g_slab_set = true;
  return NEPTUNE_HIP_OK;
}
int neptune_hip_clear_slab(void) {
  g_slab_set = false;
  g_slab_pending = nullptr;
  return NEPTUNE_HIP_OK;
}
int neptune_hip_set_slab_pending(void* event) {
  if (!g_slab_set) return NEPTUNE_HIP_EINVAL;
  g_slab_pending = event;
  return NEPTUNE_HIP_OK;
}
void* neptune_hip_get_slab_pending(void) { return g_slab_set ? g_slab_pending : nullptr; }
int neptune_hip_get_slab(int64_t out[4]) {
  if (!g_slab_set) return 0;
  for (int i = 0; i < 4; ++i) out[i] = g_slab[i];
  return 1;
}

void neptune_rt_free(void* p) {
  if (!p) return;
  if (neptune_hip_is_device_ptr(p)) NEPTUNE_HIP_CHECK(hipFree(p));
  else free(p);
}

// ---------------------------------------------------------------- apply
int neptune_hip_check_geom(const neptune_hip_apply_geom_t* g,
                           const int32_t radius[NEPTUNE_HIP_MAX_INPUTS][NEPTUNE_HIP_MAX_RANK]) {
  if (!radius) return geom_validate(g);
  return geom_check_radius(g, radius);
}

int neptune_hip_apply_builtin(int body, const neptune_hip_apply_geom_t* g, const void* const* in, void* out,
                              void* stream, const neptune_hip_launch_cfg_t* cfg) {
  if (!g || !in || !out) return NEPTUNE_HIP_EINVAL;
  if (body < 0 || body >= NEPTUNE_HIP_BODY_COUNT) return NEPTUNE_HIP_EINVAL;
  int rc = geom_validate(g);
  if (rc != NEPTUNE_HIP_OK) return rc;
  for (int k = 0; k < g->num_inputs; ++k)
    if (!in[k]) return NEPTUNE_HIP_EINVAL;
  rc = check_no_alias(g, in, out, body_elem_size(body));
  if (rc != NEPTUNE_HIP_OK) return rc;
  ensure_init();
  return body_entry(body)->apply(g, in, out, as_stream(stream), cfg);
}

// two or three chained applies of a built-in body in one pass over HBM (csrc/kernels/apply_march2.hpp)
int neptune_hip_apply_chain_builtin(int body, int applies, const neptune_hip_apply_geom_t* g, const void* const* in, void* out,
                                    void* stream, const neptune_hip_launch_cfg_t* cfg) {
  if (!g || !in || !out) return NEPTUNE_HIP_EINVAL;
  if (body < 0 || body >= NEPTUNE_HIP_BODY_COUNT) return NEPTUNE_HIP_EINVAL;
  if (applies != 2 && applies != 3) return NEPTUNE_HIP_EUNSUPPORTED;
  int rc = geom_validate(g);
  if (rc != NEPTUNE_HIP_OK) return rc;
  if (g->num_inputs != 1 || !in[0]) return NEPTUNE_HIP_EUNSUPPORTED;
  rc = check_no_alias(g, in, out, body_elem_size(body));
  if (rc != NEPTUNE_HIP_OK) return rc;
  ensure_init();
  const rtbody::Entry* e = body_entry(body);
  if (!e->chain) return NEPTUNE_HIP_EUNSUPPORTED;   // 1-D and box bodies: one launch per apply
  return e->chain(applies, g, in, out, as_stream(stream), cfg);
}
int neptune_hip_apply2_builtin(int body, const neptune_hip_apply_geom_t* g, const void* const* in, void* out,
                               void* stream, const neptune_hip_launch_cfg_t* cfg) {
  return neptune_hip_apply_chain_builtin(body, 2, g, in, out, stream, cfg);
}

// ---------------------------------------------------------------- hipGraph step loop
namespace {
// `applies` (2 or 3) chained applies in one launch when the body and the geometry allow it (else NEPTUNE_HIP_EUNSUPPORTED)
int loop_launch_chain(const LoopKey& k, int applies, int from, int to) {
  // inputs 1.. (centre-only inputs of a lowered apply: the same field at every stage) ride along unchanged
  const void* ins[NEPTUNE_HIP_MAX_INPUTS];
  for (int i = 0; i < k.g.num_inputs; ++i) ins[i] = k.in[i];
  ins[0] = k.fields[from];
  const neptune_hip_launch_cfg_t* cfg = (k.cfg.kernel || k.cfg.variant >= 0 || k.cfg.chunk || k.cfg.flags) ? &k.cfg : nullptr;
  if (cfg && (cfg->kernel == NEPTUNE_HIP_KERNEL_DIRECT || cfg->variant >= 0)) return NEPTUNE_HIP_EUNSUPPORTED;  // an explicit tile was asked for
  if (k.fn) {
    neptune_hip_apply_fn f = applies == 2 ? k.fn2 : k.fn3;
    return f ? f(&k.g, ins, k.fields[to], (void*)k.stream, cfg) : NEPTUNE_HIP_EUNSUPPORTED;
  }
  return neptune_hip_apply_chain_builtin(k.body, applies, &k.g, ins, k.fields[to], (void*)k.stream, cfg);
}
int loop_launch(const LoopKey& k, int from, int to) {
  const void* ins[NEPTUNE_HIP_MAX_INPUTS];
  for (int i = 0; i < k.g.num_inputs; ++i) ins[i] = k.in[i];
  ins[0] = k.fields[from];
  const neptune_hip_launch_cfg_t* cfg = (k.cfg.kernel || k.cfg.variant >= 0 || k.cfg.chunk || k.cfg.flags) ? &k.cfg : nullptr;
  return k.fn ? k.fn(&k.g, ins, k.fields[to], (void*)k.stream, cfg)
              : neptune_hip_apply_builtin(k.body, &k.g, ins, k.fields[to], (void*)k.stream, cfg);
}
}  // namespace

int neptune_hip_step_loop(neptune_hip_apply_fn fn, int body, const neptune_hip_apply_geom_t* g, void* const fields[2],
                          const void* const* in, int64_t steps, void* stream, const neptune_hip_launch_cfg_t* cfg) {
  return neptune_hip_step_loop_pairs(fn, nullptr, body, g, fields, in, steps, stream, cfg);
}

int neptune_hip_step_loop_pairs(neptune_hip_apply_fn fn, neptune_hip_apply_fn fn2, int body, const neptune_hip_apply_geom_t* g,
                                void* const fields[2], const void* const* in, int64_t steps, void* stream,
                                const neptune_hip_launch_cfg_t* cfg) {
  return neptune_hip_step_loop_chain(fn, fn2, nullptr, body, g, fields, in, steps, stream, cfg);
}

int neptune_hip_step_loop_chain(neptune_hip_apply_fn fn, neptune_hip_apply_fn fn2, neptune_hip_apply_fn fn3, int body,
                                const neptune_hip_apply_geom_t* g, void* const fields[2], const void* const* in, int64_t steps,
                                void* stream, const neptune_hip_launch_cfg_t* cfg) {
  if (!g || !fields || !fields[0] || !fields[1] || fields[0] == fields[1] || steps < 0) return NEPTUNE_HIP_EINVAL;
  if (g->num_inputs < 1 || g->num_inputs > NEPTUNE_HIP_MAX_INPUTS) return NEPTUNE_HIP_EINVAL;
  if (g->num_inputs > 1 && !in) return NEPTUNE_HIP_EINVAL;
  ensure_init();
  LoopKey key;
  memset(&key, 0, sizeof(key));  // padding too: keys are compared with memcmp
  key.fn = fn;
  key.fn2 = fn ? fn2 : nullptr;
  key.fn3 = fn ? fn3 : nullptr;
  key.body = fn ? -1 : body;
  key.g = *g;
  key.fields[0] = fields[0];
  key.fields[1] = fields[1];
  for (int i = 1; i < g->num_inputs; ++i) key.in[i] = in[i];
  if (cfg) key.cfg = *cfg; else key.cfg.variant = -1;
  key.stream = as_stream(stream);
  if (steps == 0) return NEPTUNE_HIP_OK;
  hipStream_t user = key.stream;
  if (!user) {
    // the legacy default stream cannot be captured: run the loop on an internal stream ordered after
    // everything already queued on the default stream, and order the default stream after the loop
    if (!g_loop_stream) {
      NEPTUNE_HIP_CHECK(hipStreamCreateWithFlags(&g_loop_stream, hipStreamNonBlocking));
      NEPTUNE_HIP_CHECK(hipEventCreateWithFlags(&g_loop_ev[0], hipEventDisableTiming));
      NEPTUNE_HIP_CHECK(hipEventCreateWithFlags(&g_loop_ev[1], hipEventDisableTiming));
    }
    NEPTUNE_HIP_CHECK(hipEventRecord(g_loop_ev[0], nullptr));
    NEPTUNE_HIP_CHECK(hipStreamWaitEvent(g_loop_stream, g_loop_ev[0], 0));
    key.stream = g_loop_stream;
  }
  auto finish = [&](int rc) {
    if (!user) {
      NEPTUNE_HIP_CHECK(hipEventRecord(g_loop_ev[1], g_loop_stream));
      NEPTUNE_HIP_CHECK(hipStreamWaitEvent(nullptr, g_loop_ev[1], 0));
    }
    return rc;
  };

  // `count` launches of `applies` chained applies each (1 = the plain apply), the first one reading fields[from0], every launch
  // moving the state to the other field.  The first launch is a plain one (it validates the request and warms the
  // launcher's one-time queries outside of stream capture); from 16 more on, a ping-pong pair of launches is captured
  // once into a hipGraph of 16 kernel nodes and replayed (graphs cached by geometry, pointers, configuration and kind), so
  // that small fields are not bound by launch overhead.  Returns NEPTUNE_HIP_EUNSUPPORTED from the FIRST launch untouched
  // (nothing has run then).
  auto run_launches = [&](int applies, int64_t count, int from0) -> int {
    if (count <= 0) return NEPTUNE_HIP_OK;
    auto launch = [&](int from, int to) { return applies == 1 ? loop_launch(key, from, to) : loop_launch_chain(key, applies, from, to); };
    const int b0 = from0, b1 = from0 ^ 1;
    int rc = launch(b0, b1);
    if (rc != NEPTUNE_HIP_OK) return rc;
    int64_t done = 1;
    constexpr int kPairs = 8;  // ping-pong pairs per graph: 16 kernel nodes amortise one graph launch
    if (count - done >= 2 * kPairs) {
      LoopKey gkey = key;
      gkey.applies = applies;
      gkey.from0 = from0;
      RuntimeState& s = rt();
      std::lock_guard<std::mutex> lk(s.mu);
      LoopGraph* slot = nullptr;
      for (auto& e : g_loops)
        if (e.exec && memcmp(&e.key, &gkey, sizeof(gkey)) == 0) slot = &e;
      if (!slot) {
        slot = &g_loops[0];
        for (auto& e : g_loops)
          if (e.stamp < slot->stamp) slot = &e;  // least recently used (empty slots have stamp 0)
        if (slot->exec) {
          (void)hipGraphExecDestroy(slot->exec);
          (void)hipGraphDestroy(slot->graph);
          slot->exec = nullptr;
          slot->graph = nullptr;
        }
        // the pair (b1 -> b0, b0 -> b1) leaves the state where it found it, so it can be replayed any number of times
        NEPTUNE_HIP_CHECK(hipStreamBeginCapture(key.stream, hipStreamCaptureModeRelaxed));
        int r1 = NEPTUNE_HIP_OK, r2 = NEPTUNE_HIP_OK;
        for (int p = 0; p < kPairs && r1 == NEPTUNE_HIP_OK && r2 == NEPTUNE_HIP_OK; ++p) {
          r1 = launch(b1, b0);
          r2 = launch(b0, b1);
        }
        hipGraph_t graph = nullptr;
        NEPTUNE_HIP_CHECK(hipStreamEndCapture(key.stream, &graph));
        if (r1 != NEPTUNE_HIP_OK || r2 != NEPTUNE_HIP_OK || !graph) {
          if (graph) (void)hipGraphDestroy(graph);
          return r1 != NEPTUNE_HIP_OK ? r1 : (r2 != NEPTUNE_HIP_OK ? r2 : NEPTUNE_HIP_EINVAL);
        }
        NEPTUNE_HIP_CHECK(hipGraphInstantiate(&slot->exec, graph, nullptr, nullptr, 0));
        slot->graph = graph;
        slot->key = gkey;
      }
      slot->stamp = ++g_loop_clock;
      for (; count - done >= 2 * kPairs; done += 2 * kPairs) NEPTUNE_HIP_CHECK(hipGraphLaunch(slot->exec, key.stream));
    }
    for (; done < count; ++done) {
      rc = launch((int)((from0 + done) % 2), (int)((from0 + done + 1) % 2));
      if (rc != NEPTUNE_HIP_OK) return rc == NEPTUNE_HIP_EUNSUPPORTED ? NEPTUNE_HIP_EINVAL : rc;
    }
    return NEPTUNE_HIP_OK;
  };

  // Several steps per pass over HBM.  Every launch -- of one, two or three chained applies -- moves the state to the other
  // field, and the newest state has to end in fields[steps % 2]:
  //   * triples: steps = 3 T + r needs T + r launches, and T + r = steps (mod 2) always: T triples, then r < 3 single steps;
  //   * pairs (when the triple entry does not exist or refuses): an EVEN number of pair launches brings the state back to
  //     fields[0], the remaining < 4 steps run as single launches.
  // The grouping does not change a bit: the same apply is evaluated, cell by cell, the same number of times on the same
  // operands.  NEPTUNE_HIP_NO_PAIRS=1 keeps one apply per pass, NEPTUNE_HIP_NO_TRIPLES=1 stops at two.
  // Chaining pays where a step is bound by HBM: a field that stays in the 256 MiB memory-side cache between steps gains
  // nothing from saved passes and loses to the chain kernels' longer dependent march (measured, profiles/r02_twostep.txt:
  // 128^3 and 1024^2 fp64 are faster one apply per launch, 256^3 and 2048^2 are faster chained: the line is 4e6 cells).
  int64_t cells = 1;
  for (int d = 0; d < g->rank; ++d) cells *= g->ub[d] > g->lb[d] ? g->ub[d] - g->lb[d] : 0;
  const char* min_cells_env = getenv("NEPTUNE_HIP_CHAIN_MIN_CELLS");
  const int64_t min_cells = min_cells_env ? atoll(min_cells_env) : (int64_t)4000000;
  const bool chain_ok = cells >= min_cells && !getenv("NEPTUNE_HIP_NO_PAIRS");
  // ... and whether chaining pays for THIS body is measured, once per (entry, geometry) and process: a Laplacian gains
  // 1.7-2.4x, a 13-point operator 1.1x on its 3x8 window, a body heavy enough to be bound by its arithmetic loses (the
  // windows overlap: every stage computes 1.3-1.8x the cells it keeps).  Each grouping the entry offers runs once to warm
  // and twice under HIP events, from fields[0] into fields[1] -- exactly what the loop's first step overwrites anyway.
  // Skipped for short loops, under stream capture and with NEPTUNE_HIP_TUNE=0 (then: the largest grouping offered).
  int best = 3;
  if (chain_ok && steps >= 8 && tune_mode() != 0) {
    struct Choice { neptune_hip_apply_fn fn, fn2, fn3; int body; neptune_hip_apply_geom_t g; int best; };
    static std::vector<Choice> choices;
    static std::mutex cmu;
    bool known = false;
    {
      std::lock_guard<std::mutex> lk(cmu);
      for (const Choice& c : choices)
        if (c.fn == key.fn && c.fn2 == key.fn2 && c.fn3 == key.fn3 && c.body == key.body && memcmp(&c.g, &key.g, sizeof(key.g)) == 0) {
          best = c.best;
          known = true;
        }
    }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(key.stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
    if (!known && cs == hipStreamCaptureStatusNone) {
      hipEvent_t e0, e1;
      NEPTUNE_HIP_CHECK(hipEventCreate(&e0));
      NEPTUNE_HIP_CHECK(hipEventCreate(&e1));
      double best_ms = -1;
      for (int applies = 1; applies <= 3; ++applies) {
        if (applies == 3 && getenv("NEPTUNE_HIP_NO_TRIPLES")) continue;
        auto one = [&] { return applies == 1 ? loop_launch(key, 0, 1) : loop_launch_chain(key, applies, 0, 1); };
        if (one() != NEPTUNE_HIP_OK) continue;              // a grouping this entry / geometry does not offer
        NEPTUNE_HIP_CHECK(hipEventRecord(e0, key.stream));
        (void)one();
        (void)one();
        NEPTUNE_HIP_CHECK(hipEventRecord(e1, key.stream));
        NEPTUNE_HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0;
        NEPTUNE_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double per_step = ms / (2.0 * applies);
        if (best_ms < 0 || per_step < 0.97 * best_ms) { best_ms = per_step; best = applies; }   // a larger grouping must win by 3 %
      }
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
      std::lock_guard<std::mutex> lk(cmu);
      choices.push_back({key.fn, key.fn2, key.fn3, key.body, key.g, best});
    }
  }
  if (steps >= 3 && chain_ok && best >= 3 && !getenv("NEPTUNE_HIP_NO_TRIPLES")) {
    const int64_t triples = steps / 3;
    const int rc3 = run_launches(3, triples, 0);
    if (rc3 == NEPTUNE_HIP_OK) return finish(run_launches(1, steps - 3 * triples, (int)(triples % 2)));
    if (rc3 != NEPTUNE_HIP_EUNSUPPORTED) return finish(rc3);
  }
  if (steps >= 4 && chain_ok && best >= 2) {
    const int64_t pairs = (steps / 2) & ~(int64_t)1;
    const int rc2 = run_launches(2, pairs, 0);
    if (rc2 == NEPTUNE_HIP_OK) return finish(run_launches(1, steps - 2 * pairs, 0));
    if (rc2 != NEPTUNE_HIP_EUNSUPPORTED) return finish(rc2);
  }
  return finish(run_launches(1, steps, 0));
}

int neptune_hip_apply_builtin_plan(int body, const neptune_hip_apply_geom_t* g, const void* const* in,
                                   const void* out, const neptune_hip_launch_cfg_t* cfg) {
  if (!g || !in || !out) return NEPTUNE_HIP_EINVAL;
  const rtbody::Entry* e = body_entry(body);
  return e ? e->plan(g, in, out, cfg) : NEPTUNE_HIP_EINVAL;
}

int neptune_hip_apply_builtin_variant(int body, const neptune_hip_apply_geom_t* g, const neptune_hip_launch_cfg_t* cfg) {
  if (!g) return NEPTUNE_HIP_EINVAL;
  const rtbody::Entry* e = body_entry(body);
  return e ? e->variant(g, cfg) : NEPTUNE_HIP_EINVAL;
}

const char* neptune_hip_kernel_name(int kernel) {
  switch (kernel) {
    case NEPTUNE_HIP_KERNEL_DIRECT: return "neptune_apply_direct";
    case NEPTUNE_HIP_KERNEL_MARCH: return "neptune_apply_march";
  }
  return "";
}
int neptune_hip_march_variant_count(int rank) { return march_variant_count(rank); }
const char* neptune_hip_march_variant_name(int rank, int variant) {
  const MarchVariant* v = march_variant(rank, variant);
  return v ? v->name : "";
}

// ---------------------------------------------------------------- store
int neptune_hip_store_full(int dtype, const void* src, void* dst, int64_t count, void* stream) {
  if (!src || !dst || count < 0) return NEPTUNE_HIP_EINVAL;
  if (dtype != NEPTUNE_HIP_F64 && dtype != NEPTUNE_HIP_F32) return NEPTUNE_HIP_EINVAL;
  const size_t bytes = (size_t)count * (dtype == NEPTUNE_HIP_F64 ? 8 : 4);
  if (bytes == 0 || src == dst) return NEPTUNE_HIP_OK;
  // 16-byte-aligned device buffers: the streaming copy kernel (one 16-byte load per lane, non-temporal store: 6.4 TB/s on a
  // 1024^3 fp64 field against 5.1 for hipMemcpyAsync, bench.py roofline.copy_ceiling); anything else, and the last few
  // bytes, through the runtime's copy
  const bool aligned = ((uintptr_t)src | (uintptr_t)dst) % 16 == 0;
  const int64_t n16 = aligned ? (int64_t)(bytes / 16) : 0;
  if (n16 > 0) {
    ensure_init();
    const int64_t blocks = (n16 + 255) / 256;
    if (blocks <= 0x7fffffffLL) {
      hipLaunchKernelGGL((neptune_copy16_unrolled<1, false, true>), dim3((uint32_t)blocks), dim3(256), 0, as_stream(stream),
                         (const u32x4*)src, (u32x4*)dst, n16);
      NEPTUNE_HIP_CHECK(hipGetLastError());
      const size_t done = (size_t)n16 * 16;
      if (done < bytes)
        NEPTUNE_HIP_CHECK(hipMemcpyAsync((char*)dst + done, (const char*)src + done, bytes - done, hipMemcpyDeviceToDevice, as_stream(stream)));
      return NEPTUNE_HIP_OK;
    }
  }
  NEPTUNE_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
  return NEPTUNE_HIP_OK;
}

int neptune_hip_store_box(int dtype, int rank, const void* src, const int64_t* src_lb, const int64_t* src_ub,
                          void* dst, const int64_t* dst_lb, const int64_t* dst_ub, const int64_t* lb,
                          const int64_t* ub, void* stream) {
  if (!src || !dst || !src_lb || !src_ub || !dst_lb || !dst_ub || !lb || !ub) return NEPTUNE_HIP_EINVAL;
  if (rank < 1 || rank > kMaxRank) return NEPTUNE_HIP_EINVAL;
  if (dtype != NEPTUNE_HIP_F64 && dtype != NEPTUNE_HIP_F32) return NEPTUNE_HIP_EINVAL;
  int64_t ext[3], soff[3], doff[3], ss[3], ds[3];
  for (int d = 0; d < rank; ++d) {
    ext[d] = ub[d] - lb[d];
    soff[d] = lb[d] - src_lb[d];
    doff[d] = lb[d] - dst_lb[d];
    ss[d] = src_ub[d] - src_lb[d];
    ds[d] = dst_ub[d] - dst_lb[d];
    if (ext[d] < 0 || ss[d] <= 0 || ds[d] <= 0) return NEPTUNE_HIP_EINVAL;
    // the subviews of the reference (DataflowLowering.cpp:212-215) must lie inside their buffers
    if (ext[d] > 0 && (soff[d] < 0 || soff[d] + ext[d] > ss[d] || doff[d] < 0 || doff[d] + ext[d] > ds[d]))
      return NEPTUNE_HIP_EOOB;
  }
  BoxCopyParams P;
  auto fill = [&](const int64_t* a, int64_t* o, int64_t f) {
    o[0] = o[1] = o[2] = f;
    if (rank == 3) { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; }
    else if (rank == 2) { o[0] = a[0]; o[2] = a[1]; }
    else { o[2] = a[0]; }
  };
  fill(ext, P.ext, 1);
  fill(soff, P.soff, 0);
  fill(doff, P.doff, 0);
  fill(ss, P.sshape, 1);
  fill(ds, P.dshape, 1);
  const int64_t total = P.ext[0] * P.ext[1] * P.ext[2];
  if (total == 0) return NEPTUNE_HIP_OK;
  // one workgroup per chunk of 256 lanes x 16 bytes of one row of the box (see the kernel)
  const int64_t chunk_cells = 256 * (dtype == NEPTUNE_HIP_F64 ? 2 : 4);
  const int64_t blocks = P.ext[0] * P.ext[1] * ((P.ext[2] + chunk_cells - 1) / chunk_cells);
  if (blocks > 0x7fffffffLL) return NEPTUNE_HIP_EUNSUPPORTED;
  ensure_init();
  if (dtype == NEPTUNE_HIP_F64)
    hipLaunchKernelGGL(neptune_store_box<double>, grid_for_blocks(blocks), dim3(256), 0, as_stream(stream),
                       (const double*)src, (double*)dst, P);
  else
    hipLaunchKernelGGL(neptune_store_box<float>, grid_for_blocks(blocks), dim3(256), 0, as_stream(stream),
                       (const float*)src, (float*)dst, P);
  NEPTUNE_HIP_CHECK(hipGetLastError());
  return NEPTUNE_HIP_OK;
}

// ---------------------------------------------------------------- reduce
void* neptune_hip_reduce_workspace(void) {
  ensure_init();
  return rt().reduce_ws;
}

int neptune_hip_reduce_sum(int dtype, int rank, const void* src, const int64_t* src_lb, const int64_t* src_ub,
                           const int64_t* lb, const int64_t* ub, double* result, void* stream) {
  if (!src || !src_lb || !src_ub || !result) return NEPTUNE_HIP_EINVAL;
  if (rank < 1 || rank > kMaxRank) return NEPTUNE_HIP_EINVAL;
  if (dtype != NEPTUNE_HIP_F64 && dtype != NEPTUNE_HIP_F32) return NEPTUNE_HIP_EINVAL;
  int64_t ext[3], off[3], shp[3];
  bool whole = true;
  int64_t total = 1, count = 1;
  for (int d = 0; d < rank; ++d) {
    shp[d] = src_ub[d] - src_lb[d];
    if (shp[d] <= 0) return NEPTUNE_HIP_EINVAL;
    const int64_t l = lb ? lb[d] : src_lb[d], u = ub ? ub[d] : src_ub[d];
    ext[d] = u - l;
    off[d] = l - src_lb[d];
    if (ext[d] < 0) return NEPTUNE_HIP_EINVAL;
    if (ext[d] > 0 && (off[d] < 0 || off[d] + ext[d] > shp[d])) return NEPTUNE_HIP_EOOB;  // memref.load out of range
    whole = whole && off[d] == 0 && ext[d] == shp[d] && ((uintptr_t)src % 16 == 0);
    total *= ext[d];
    count *= shp[d];
  }
  if (total == 0) {  // empty domain: the reference's loop never runs, the accumulator stays 0
    *result = 0.0;
    return NEPTUNE_HIP_OK;
  }
  ensure_init();
  RuntimeState& s = rt();
  hipStream_t st = as_stream(stream);
  int blocks = (int)((total + 255) / 256 < kReduceBlocks ? (total + 255) / 256 : kReduceBlocks);
  if (!whole) {  // box kernel: one unit of work = 4 row chunks of 256 lanes x 16 bytes
    const int64_t cells = 256 * (dtype == NEPTUNE_HIP_F64 ? 2 : 4);
    const int64_t last = ext[rank - 1];
    const int64_t trips = ((total / (last ? last : 1)) * ((last + cells - 1) / cells) + 3) / 4;
    blocks = (int)(trips < kReduceBlocks ? (trips < 1 ? 1 : trips) : kReduceBlocks);
  }
  ReduceBoxParams P;
  auto fill = [&](const int64_t* a, int64_t* o, int64_t f) {
    o[0] = o[1] = o[2] = f;
    if (rank == 3) { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; }
    else if (rank == 2) { o[0] = a[0]; o[2] = a[1]; }
    else { o[2] = a[0]; }
  };
  fill(ext, P.ext, 1);
  fill(off, P.off, 0);
  fill(shp, P.shape, 1);
  if (dtype == NEPTUNE_HIP_F64) {
    double* part = (double*)s.reduce_ws;
    if (whole) hipLaunchKernelGGL(neptune_reduce_partial_flat<double>, dim3(blocks), dim3(256), 0, st, (const double*)src, count, part);
    else hipLaunchKernelGGL(neptune_reduce_partial_box<double>, dim3(blocks), dim3(256), 0, st, (const double*)src, P, part);
    hipLaunchKernelGGL(neptune_reduce_final<double>, dim3(1), dim3(256), 0, st, part, blocks, part + kReduceBlocks);
    NEPTUNE_HIP_CHECK(hipGetLastError());
    double h = 0;
    NEPTUNE_HIP_CHECK(hipMemcpyAsync(&h, part + kReduceBlocks, sizeof(double), hipMemcpyDeviceToHost, st));
    NEPTUNE_HIP_CHECK(hipStreamSynchronize(st));
    *result = h;
  } else {
    float* part = (float*)s.reduce_ws;
    if (whole) hipLaunchKernelGGL(neptune_reduce_partial_flat<float>, dim3(blocks), dim3(256), 0, st, (const float*)src, count, part);
    else hipLaunchKernelGGL(neptune_reduce_partial_box<float>, dim3(blocks), dim3(256), 0, st, (const float*)src, P, part);
    hipLaunchKernelGGL(neptune_reduce_final<float>, dim3(1), dim3(256), 0, st, part, blocks, part + kReduceBlocks);
    NEPTUNE_HIP_CHECK(hipGetLastError());
    float h = 0;
    NEPTUNE_HIP_CHECK(hipMemcpyAsync(&h, part + kReduceBlocks, sizeof(float), hipMemcpyDeviceToHost, st));
    NEPTUNE_HIP_CHECK(hipStreamSynchronize(st));
    *result = (double)h;
  }
  return NEPTUNE_HIP_OK;
}

// ---------------------------------------------------------------- Krylov vector updates
int neptune_hip_axpy(int dtype, int64_t n, double a, const void* x, void* y, void* stream) {
  return vec_update<false>(dtype, n, a, x, y, stream);
}
int neptune_hip_xpay(int dtype, int64_t n, const void* x, double a, void* y, void* stream) {
  return vec_update<true>(dtype, n, a, x, y, stream);
}

// ---------------------------------------------------------------- helpers
int neptune_hip_fill_hash(int dtype, void* dst, int64_t count, int64_t index_offset, uint64_t seed,
                          void* stream) {
  if (!dst || count < 0) return NEPTUNE_HIP_EINVAL;
  if (dtype != NEPTUNE_HIP_F64 && dtype != NEPTUNE_HIP_F32) return NEPTUNE_HIP_EINVAL;
  if (count == 0) return NEPTUNE_HIP_OK;
  ensure_init();
  const int64_t want = (count + 255) / 256;
  const uint32_t blocks = (uint32_t)(want < 8192 ? want : 8192);
  if (dtype == NEPTUNE_HIP_F64)
    hipLaunchKernelGGL(neptune_fill_hash<double>, dim3(blocks), dim3(256), 0, as_stream(stream), (double*)dst,
                       count, index_offset, seed);
  else
    hipLaunchKernelGGL(neptune_fill_hash<float>, dim3(blocks), dim3(256), 0, as_stream(stream), (float*)dst,
                       count, index_offset, seed);
  NEPTUNE_HIP_CHECK(hipGetLastError());
  return NEPTUNE_HIP_OK;
}

double neptune_hip_hash_value(int dtype, int64_t index, uint64_t seed) {
  return dtype == NEPTUNE_HIP_F32 ? (double)hash_f32(index, seed) : hash_f64(index, seed);
}

int64_t neptune_hip_count_mismatch(int dtype, const void* a, const void* b, int64_t count, void* stream) {
  if (!a || !b || count < 0) return -1;
  if (dtype != NEPTUNE_HIP_F64 && dtype != NEPTUNE_HIP_F32) return -1;
  if (count == 0) return 0;
  ensure_init();
  RuntimeState& s = rt();
  hipStream_t st = as_stream(stream);
  NEPTUNE_HIP_CHECK(hipMemsetAsync(s.counter, 0, sizeof(unsigned long long), st));
  const int64_t want = (count + 255) / 256;
  const uint32_t blocks = (uint32_t)(want < 8192 ? want : 8192);
  if (dtype == NEPTUNE_HIP_F64)
    hipLaunchKernelGGL(neptune_count_mismatch<unsigned long long>, dim3(blocks), dim3(256), 0, st,
                       (const unsigned long long*)a, (const unsigned long long*)b, count, s.counter);
  else
    hipLaunchKernelGGL(neptune_count_mismatch<unsigned int>, dim3(blocks), dim3(256), 0, st,
                       (const unsigned int*)a, (const unsigned int*)b, count, s.counter);
  NEPTUNE_HIP_CHECK(hipGetLastError());
  unsigned long long host = 0;
  NEPTUNE_HIP_CHECK(hipMemcpyAsync(&host, s.counter, sizeof(host), hipMemcpyDeviceToHost, st));
  NEPTUNE_HIP_CHECK(hipStreamSynchronize(st));
  return (int64_t)host;
}

double neptune_hip_time_apply_builtin(int body, const neptune_hip_apply_geom_t* g, const void* const* in,
                                      void* out, void* stream, const neptune_hip_launch_cfg_t* cfg, int warmup,
                                      int reps) {
  return time_launches([&](const neptune_hip_launch_cfg_t* c) { return neptune_hip_apply_builtin(body, g, in, out, stream, c); },
                       as_stream(stream), cfg, warmup, reps);
}

double neptune_hip_time_apply_fn(neptune_hip_apply_fn fn, const neptune_hip_apply_geom_t* g, const void* const* in,
                                 void* out, void* stream, const neptune_hip_launch_cfg_t* cfg, int warmup, int reps) {
  if (!fn) return (double)NEPTUNE_HIP_EINVAL;
  return time_launches([&](const neptune_hip_launch_cfg_t* c) { return fn(g, in, out, stream, c); }, as_stream(stream), cfg,
                       warmup, reps);
}

int neptune_hip_autotune_builtin(int body, const neptune_hip_apply_geom_t* g, const void* const* in, void* out,
                                 void* stream, int reps, neptune_hip_launch_cfg_t* best, double* best_ms) {
  if (!g || !in || !out || !best) return NEPTUNE_HIP_EINVAL;
  neptune_hip_launch_cfg_t probe = {NEPTUNE_HIP_KERNEL_AUTO, -1, 0, 0};
  const int planned = neptune_hip_apply_builtin_plan(body, g, in, out, &probe);
  if (planned < 0) return planned;
  return autotune_launches([&](const neptune_hip_launch_cfg_t* c) { return neptune_hip_apply_builtin(body, g, in, out, stream, c); },
                           planned == NEPTUNE_HIP_KERNEL_MARCH, g->rank, march_variant_count(g->rank), as_stream(stream), reps,
                           best, best_ms);
}

int neptune_hip_autotune_fn(neptune_hip_apply_fn fn, int num_variants, const neptune_hip_apply_geom_t* g,
                            const void* const* in, void* out, void* stream, int reps, neptune_hip_launch_cfg_t* best,
                            double* best_ms) {
  if (!fn || !g || !in || !out || !best) return NEPTUNE_HIP_EINVAL;
  if (g->rank < 1 || g->rank > kMaxRank) return NEPTUNE_HIP_EINVAL;
  // whether the automatic launch is a march launch is the module's business; a forced march tile that cannot take
  // the geometry is rejected by the entry itself (negative return) and simply never becomes the best
  const int nv = num_variants < 0 ? 0 : (num_variants < march_variant_count(g->rank) ? num_variants : march_variant_count(g->rank));
  return autotune_launches([&](const neptune_hip_launch_cfg_t* c) { return fn(g, in, out, stream, c); }, nv > 0, g->rank, nv,
                           as_stream(stream), reps, best, best_ms);
}

double neptune_hip_time_copy(void* dst, const void* src, size_t bytes, void* stream, int mode, int warmup,
                             int reps) {
  if (!dst || !src || reps <= 0 || bytes % 16 != 0) return -1.0;
  ensure_init();
  hipStream_t st = as_stream(stream);
  const int64_t n16 = (int64_t)(bytes / 16);
  auto blocks_for = [&](int U) { return (uint32_t)((n16 + 256 * U - 1) / (256 * U)); };
  auto launch = [&] {
    const u32x4* s4 = (const u32x4*)src;
    u32x4* d4 = (u32x4*)dst;
    switch (mode) {
      default:
      case 0:  // grid-stride, 8 workgroups per CU
        hipLaunchKernelGGL(neptune_copy16, dim3(256 * 8), dim3(256), 0, st, (const uint4*)src, (uint4*)dst, n16);
        break;
      case 1: hipLaunchKernelGGL((neptune_copy16_unrolled<4, false, false>), dim3(blocks_for(4)), dim3(256), 0, st, s4, d4, n16); break;
      case 2: hipLaunchKernelGGL((neptune_copy16_unrolled<4, false, true>), dim3(blocks_for(4)), dim3(256), 0, st, s4, d4, n16); break;
      case 3: hipLaunchKernelGGL((neptune_copy16_unrolled<4, true, true>), dim3(blocks_for(4)), dim3(256), 0, st, s4, d4, n16); break;
      case 4: hipLaunchKernelGGL((neptune_copy16_unrolled<8, false, true>), dim3(blocks_for(8)), dim3(256), 0, st, s4, d4, n16); break;
      case 5: hipLaunchKernelGGL((neptune_copy16_unrolled<1, false, true>), dim3(blocks_for(1)), dim3(256), 0, st, s4, d4, n16); break;
      case 6: hipLaunchKernelGGL((neptune_copy16_unrolled<2, false, true>), dim3(blocks_for(2)), dim3(256), 0, st, s4, d4, n16); break;
    }
  };
  for (int i = 0; i < warmup; ++i) launch();
  hipEvent_t e0, e1;
  NEPTUNE_HIP_CHECK(hipEventCreate(&e0));
  NEPTUNE_HIP_CHECK(hipEventCreate(&e1));
  NEPTUNE_HIP_CHECK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) launch();
  NEPTUNE_HIP_CHECK(hipEventRecord(e1, st));
  NEPTUNE_HIP_CHECK(hipEventSynchronize(e1));
  NEPTUNE_HIP_CHECK(hipGetLastError());
  float ms = 0.f;
  NEPTUNE_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  NEPTUNE_HIP_CHECK(hipEventDestroy(e0));
  NEPTUNE_HIP_CHECK(hipEventDestroy(e1));
  return (double)ms / reps;
}
int neptune_hip_copy_mode_count(void) { return 7; }

void* neptune_hip_event_create(void) {
  ensure_init();
  hipEvent_t e;
  NEPTUNE_HIP_CHECK(hipEventCreate(&e));
  return (void*)e;
}
void neptune_hip_event_destroy(void* ev) {
  if (ev) NEPTUNE_HIP_CHECK(hipEventDestroy((hipEvent_t)ev));
}
void neptune_hip_event_record(void* ev, void* stream) {
  NEPTUNE_HIP_CHECK(hipEventRecord((hipEvent_t)ev, as_stream(stream)));
}
void neptune_hip_event_sync(void* ev) { NEPTUNE_HIP_CHECK(hipEventSynchronize((hipEvent_t)ev)); }
double neptune_hip_event_elapsed_ms(void* start, void* stop) {
  float ms = 0.f;
  NEPTUNE_HIP_CHECK(hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop));
  return (double)ms;
}
void neptune_hip_stream_wait_event(void* stream, void* ev) {
  NEPTUNE_HIP_CHECK(hipStreamWaitEvent(as_stream(stream), (hipEvent_t)ev, 0));
}

}  // extern "C"

// lowered_runtime.hpp -- host-side support code included by every module the lowering emits.
//
// A lowered module exports one extern "C" symbol per func.func / opdef with the reference's
// calling convention (expanded memref arguments, memref struct result; reference:
// test/smoke_tests/smoke_apply.sh:39-50, include/Runtime/PETSc/NeptunePETScRuntime.h:22-42).  The
// helpers here implement what the reference's lowering leaves to malloc/memcpy:
//
//   * residency : a memref argument may point to host or to device memory.  Host buffers get a
//     device shadow (H2D once at entry, D2H at exit if a neptune_ir.store dirtied it), so existing
//     host drivers and the PETSc MatMult thunk (NeptunePETScRuntime.cpp:182-230) keep working;
//     device buffers are used in place, which is the fast path (nothing crosses PCIe).
//   * ownership : an apply result is allocated by the callee and released by the caller
//     (DataflowLowering.cpp:281; NeptunePETScRuntime.cpp:219-221 `free(yout.allocated)`).  If any
//     argument was host memory the result is returned in malloc'ed host memory so `free()` works;
//     otherwise it is device memory to be released with neptune_rt_free().
//   * aliasing  : wrap / unwrap / load are aliases (DataflowLowering.cpp:131-159); `store
//     apply(load f) to f` must behave as if the apply had a private result (:281 + :176-179), so an
//     apply only writes straight into a destination field when that field aliases none of its
//     inputs -- otherwise it goes through a temporary and a device copy.
//   * errors    : "[NeptuneRT][HIP] ..." on stderr + abort(), as the reference runtime does.
//   * slabs     : while neptune_hip_set_slab() is in force (one process per GPU) the module's global
//     boxes are read as this rank's local boxes and every bounds attribute is clipped to the owned
//     planes; see include/neptune_hip.h and neptune_hip/slab.py ShardedModule.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../../include/neptune_hip.h"
#include "../kernels/apply_launch.hpp"
#include "../kernels/apply_march2.hpp"
#include "../kernels/apply_nd.hpp"
#include "../kernels/body_ops.hpp"
#include "../kernels/reduce_apply.hpp"

namespace neptune_hip {
namespace lowered {

// fields / temps / memrefs of rank 1..6; kernels take rank 1..3, the leading dimensions of a wider apply are peeled off on
// the host (run_apply_batched)
constexpr int kMaxBoxRank = 6;
struct Box {
  int rank;
  int64_t lb[kMaxBoxRank], ub[kMaxBoxRank];
  int64_t count() const {
    int64_t n = 1;
    for (int d = 0; d < rank; ++d) n *= (ub[d] - lb[d]);
    return n;
  }
  bool same_shape(const Box& o) const {
    if (rank != o.rank) return false;
    for (int d = 0; d < rank; ++d)
      if (ub[d] - lb[d] != o.ub[d] - o.lb[d]) return false;
    return true;
  }
};

// a memref / field / temp SSA value at run time
struct Val {
  void* dev = nullptr;    // device address of element 0 (dense row-major)
  int64_t count = 0;      // elements
  int esize = 8;
  Box box{};              // logical box; memref values use lb = 0
  int shadow = -1;        // index of the host-backed argument this value aliases, or -1
  bool stale_ghosts = false;  // slab mode: produced inside this call, its ghost planes hold no valid data
};

struct HostArg {          // a memref argument that arrived in host memory
  void* host;             // first element (aligned + offset)
  void* dev;              // its device shadow
  size_t bytes;
  bool dirty;             // device copy is newer (a store wrote it): flush at exit
};

[[noreturn]] inline void die(const char* fn, const char* what) {
  fprintf(stderr, "[NeptuneRT][HIP] %s: %s\n", fn, what);
  abort();
}

class Scope {
 public:
  explicit Scope(const char* fn) : fn_(fn) {
    neptune_hip_init_default();
    slab_on_ = neptune_hip_get_slab(slab_) == 1;
    pending_ = slab_on_ ? neptune_hip_get_slab_pending() : nullptr;
  }
  // ---- a halo exchange of this call's inputs still in flight on another stream (neptune_hip_set_slab_pending) ----
  bool pending() const { return pending_ != nullptr; }
  // make this function's stream wait for it; afterwards every ghost plane of the inputs is valid
  void wait_pending() {
    if (!pending_) return;
    NEPTUNE_HIP_CHECK(hipStreamWaitEvent(stream(), static_cast<hipEvent_t>(pending_), 0));
    pending_ = nullptr;
    neptune_hip_set_slab_pending(nullptr);   // consumed: later calls under the same slab view do not wait again
  }
  int64_t ghost_lo() const { return slab_on_ ? slab_[2] : 0; }
  int64_t ghost_hi() const { return slab_on_ ? slab_[3] : 0; }
  // ---- slab view -------------------------------------------------------------------------
  bool slab() const { return slab_on_; }
  bool has_ghosts() const { return slab_on_ && (slab_[2] > 0 || slab_[3] > 0); }
  // a declared (global) logical box -> what this rank holds of it
  Box local_box(const Box& b) const {
    Box r = b;
    if (!slab_on_) return r;
    const int64_t lo = slab_[0] - slab_[2], hi = slab_[1] + slab_[3];
    if (r.lb[0] < lo) r.lb[0] = lo;
    if (r.ub[0] > hi) r.ub[0] = hi;
    if (r.ub[0] < r.lb[0]) r.ub[0] = r.lb[0];
    return r;
  }
  // loop bounds with lb > ub along a dimension: the reference's scf.for nest (DataflowLowering.cpp:289-310 for
  // apply, :604-611 for reduce) makes zero trips, i.e. the box is empty
  static Box zero_trip(const Box& b) {
    Box r = b;
    for (int d = 0; d < r.rank; ++d)
      if (r.ub[d] < r.lb[d]) r.ub[d] = r.lb[d];
    return r;
  }
  // a bounds attribute -> the part this rank computes: its owned planes, or (with_ghosts) all the
  // planes it holds
  Box owned_bounds(const Box& b, bool with_ghosts = false) const {
    Box r = zero_trip(b);
    if (!slab_on_) return r;
    const int64_t lo = slab_[0] - (with_ghosts ? slab_[2] : 0), hi = slab_[1] + (with_ghosts ? slab_[3] : 0);
    if (r.lb[0] < lo) r.lb[0] = lo;
    if (r.ub[0] > hi) r.ub[0] = hi;
    if (r.ub[0] < r.lb[0]) r.ub[0] = r.lb[0];
    return r;
  }
  ~Scope() {
    // every exit path has synchronised the stream (finish / export_result), so the blocks are idle
    for (auto& b : owned_) neptune_hip_pool_release(b.p, b.bytes);
  }
  const char* name() const { return fn_; }
  bool host_mode() const { return !host_args_.empty(); }
  hipStream_t stream() const { return nullptr; }

  // ---- arguments -----------------------------------------------------------------------
  // memref argument in the expanded ABI; sizes/strides have `rank` entries
  Val bind_memref(int rank, int esize, void* allocated, void* aligned, int64_t offset, const int64_t* sizes,
                  const int64_t* strides) {
    (void)allocated;
    if (!aligned) die(fn_, "null memref argument");
    Val v;
    v.esize = esize;
    v.box.rank = rank;
    int64_t expect = 1;
    v.count = 1;
    for (int d = rank - 1; d >= 0; --d) {
      if (sizes[d] <= 0) die(fn_, "memref argument with a non-positive extent");
      if (sizes[d] != 1 && strides[d] != expect)
        die(fn_, "memref argument is not dense row-major (the reference's memref.cast to the static field type "
                 "requires the identity layout)");
      expect *= sizes[d];
      v.count *= sizes[d];
      v.box.lb[d] = 0;
      v.box.ub[d] = sizes[d];
    }
    char* first = static_cast<char*>(aligned) + offset * esize;
    if (neptune_hip_is_device_ptr(first)) {
      v.dev = first;
    } else {
      HostArg h;
      h.host = first;
      h.bytes = (size_t)v.count * esize;
      h.dirty = false;
      h.dev = neptune_hip_pool_alloc(h.bytes);
      owned_.push_back({h.dev, h.bytes});
      NEPTUNE_HIP_CHECK(hipMemcpy(h.dev, h.host, h.bytes, hipMemcpyHostToDevice));
      v.dev = h.dev;
      v.shadow = (int)host_args_.size();
      host_args_.push_back(h);
    }
    return v;
  }

  // wrap / unwrap / load: same buffer, new logical box (memref.cast ?->static must be valid)
  Val alias(const Val& src, const Box& declared, const char* op) {
    Box box = local_box(declared);
    if (slab_on_ && !strcmp(op, "neptune_ir.unwrap")) {
      // unwrap -> memref: the zero-based box of whatever the field holds locally
      box = src.box;
      for (int d = 0; d < box.rank; ++d) { box.ub[d] -= box.lb[d]; box.lb[d] = 0; }
    }
    if (box.count() != src.count || !box_matches(src, box)) {
      fprintf(stderr, "[NeptuneRT][HIP] %s: %s: buffer shape does not match the declared field/temp bounds\n", fn_, op);
      abort();
    }
    Val v = src;
    v.box = box;
    return v;
  }

  // ---- results -------------------------------------------------------------------------
  Val alloc(const Box& box, int esize) {
    Val v;
    v.esize = esize;
    v.box = box;
    v.count = box.count();
    size_t bytes = (size_t)v.count * esize;
    v.dev = neptune_hip_pool_alloc(bytes);
    owned_.push_back({v.dev, bytes});
    return v;
  }
  void mark_dirty(const Val& v) {
    if (v.shadow >= 0) host_args_[v.shadow].dirty = true;
  }
  // hand a value to the caller: its device buffer leaves this scope's ownership
  void* release(const Val& v) {
    for (size_t i = 0; i < owned_.size(); ++i)
      if (owned_[i].p == v.dev) {
        owned_.erase(owned_.begin() + i);
        return v.dev;
      }
    return nullptr;  // not owned here (an argument or an alias of one)
  }
  // NEPTUNE_HIP_ASYNC=1: a call whose buffers are all device memory returns without waiting for its kernels;
  // results are then ordered on the default stream like any other work queued there (torch's default stream on
  // ROCm), which removes the per-call synchronisation for device-resident time loops.  Temporaries go back to the
  // block pool while still in flight: safe, because every lowered function and therefore every reuse of a pooled
  // block is queued on the same stream.  Host buffers and scalar results always synchronise.
  static bool async_mode() {
    static const bool on = [] {
      const char* e = getenv("NEPTUNE_HIP_ASYNC");
      return e && *e && *e != '0';
    }();
    return on;
  }
  // flush dirty host shadows; called by the exported wrapper before returning
  void finish() {
    wait_pending();   // a function that never needed the ghost planes still returns with the exchange ordered before it
    if (async_mode() && !host_mode()) return;
    NEPTUNE_HIP_CHECK(hipStreamSynchronize(stream()));
    for (auto& h : host_args_)
      if (h.dirty) NEPTUNE_HIP_CHECK(hipMemcpy(h.host, h.dev, h.bytes, hipMemcpyDeviceToHost));
  }
  // result buffer for the caller: host malloc in host mode (caller free()s it), else the device
  // buffer itself (caller neptune_rt_free()s it)
  void* export_result(const Val& v) {
    wait_pending();
    if (!(async_mode() && !host_mode())) NEPTUNE_HIP_CHECK(hipStreamSynchronize(stream()));
    void* owned = release(v);
    if (host_mode()) {
      size_t bytes = (size_t)v.count * v.esize;
      void* h = malloc(bytes ? bytes : 1);
      if (!h) die(fn_, "malloc of the result failed");
      NEPTUNE_HIP_CHECK(hipMemcpy(h, v.dev, bytes, hipMemcpyDeviceToHost));
      if (owned) neptune_hip_pool_release(owned, bytes);
      return h;
    }
    if (owned) return owned;
    // returning an alias of an argument that is not ours to give away: hand out a private copy so
    // the callee-allocates / caller-frees rule still holds
    void* d = nullptr;
    size_t bytes = (size_t)v.count * v.esize;
    d = neptune_hip_pool_alloc(bytes);
    NEPTUNE_HIP_CHECK(hipMemcpy(d, v.dev, bytes, hipMemcpyDeviceToDevice));
    return d;
  }

 private:
  static void neptune_hip_init_default() {
    static bool once = false;
    if (!once) {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess) die("init", "no HIP device available");
      neptune_hip_init(dev);
      once = true;
    }
  }
  static bool box_matches(const Val& v, const Box& b) {
    if (v.box.rank != b.rank) return false;
    return v.box.same_shape(b);
  }
  const char* fn_;
  void* pending_ = nullptr;
  bool slab_on_ = false;
  int64_t slab_[4] = {0, 0, 0, 0};
  struct Owned { void* p; size_t bytes; };
  std::vector<Owned> owned_;
  std::vector<HostArg> host_args_;
};

inline bool overlaps(const Val& a, const Val& b) {
  const uintptr_t x = (uintptr_t)a.dev, y = (uintptr_t)b.dev;
  const uintptr_t na = (uintptr_t)a.count * a.esize, nb = (uintptr_t)b.count * b.esize;
  return x < y + nb && y < x + na;
}

inline void fill_geom(neptune_hip_apply_geom_t& g, const Box& out, const Box& bounds, const Val* const* in, int nin) {
  memset(&g, 0, sizeof(g));
  g.rank = out.rank;
  g.num_inputs = nin;
  for (int d = 0; d < out.rank; ++d) {
    g.out_lb[d] = out.lb[d];
    g.out_ub[d] = out.ub[d];
    g.lb[d] = bounds.lb[d];
    g.ub[d] = bounds.ub[d];
    g.region_lb[d] = 0;
    g.region_ub[d] = out.ub[d] - out.lb[d];
    for (int k = 0; k < nin; ++k) {
      g.in_lb[k][d] = in[k]->box.lb[d];
      g.in_ub[k][d] = in[k]->box.ub[d];
    }
  }
}

// One neptune_ir.apply.  `dest`: a field the single consumer (a whole-buffer store) will copy the
// result into; when it aliases no input the kernel writes there directly and the copy is elided.
// `top_radius`: max |offset| of the UNCONDITIONAL accesses (those not nested under scf.if) -- the
// ones that certainly execute for every in-bounds point and must therefore stay inside their
// input's box (out of bounds = undefined behaviour in the reference, rejected here).
// Tuning/testing override for every apply of a lowered module, read at each launch:
//   NEPTUNE_HIP_KERNEL=direct|direct-flat|march   NEPTUNE_HIP_VARIANT=<tile index>   NEPTUNE_HIP_CHUNK=<planes>
// A forced march kernel that cannot take the launch (narrow or unaligned rows) is an error, as in
// neptune_hip_apply_builtin.  Unset: the automatic choice.
inline const neptune_hip_launch_cfg_t* launch_override() {
  static thread_local neptune_hip_launch_cfg_t cfg;
  const char* k = getenv("NEPTUNE_HIP_KERNEL");
  const char* v = getenv("NEPTUNE_HIP_VARIANT");
  const char* c = getenv("NEPTUNE_HIP_CHUNK");
  if (!k && !v && !c) return nullptr;
  const bool flat = k && !strcmp(k, "direct-flat");
  cfg.kernel = !k ? NEPTUNE_HIP_KERNEL_AUTO
                  : ((flat || !strcmp(k, "direct")) ? NEPTUNE_HIP_KERNEL_DIRECT
                                                    : (!strcmp(k, "march") ? NEPTUNE_HIP_KERNEL_MARCH : NEPTUNE_HIP_KERNEL_AUTO));
  cfg.variant = v ? atoi(v) : -1;
  cfg.chunk = c ? atoi(c) : 0;
  cfg.flags = flat ? NEPTUNE_HIP_FLAG_DIRECT_FLAT : 0;
  return &cfg;
}

// `halo0`: max |offset| along dim 0 over ALL accesses of the body (slab mode: how far the apply
// reaches into its inputs' ghost planes).
template <class Body, class T, int RANK, int NIN, class FP>
inline Val run_apply(Scope& sc, const Body& body, const Box& result_decl, const Box& bounds_decl, const Val* const* in,
                     const neptune_hip::Reach& top_radius, const Val* dest, int halo0 = 0) {
  // slab mode: an apply that stays within its plane (halo0 == 0) and reads only values whose ghost
  // planes are good is computed on the ghost planes too, so a stencil apply may follow it without an
  // exchange; any other result has stale ghosts
  bool inputs_fresh = true;
  for (int k = 0; k < NIN; ++k) inputs_fresh = inputs_fresh && !in[k]->stale_ghosts;
  if (sc.has_ghosts() && halo0 > 0 && !inputs_fresh)
    die(sc.name(), "slab mode: neptune_ir.apply reads neighbouring planes of a value computed inside this call; "
                   "its ghost planes would need a halo exchange in the middle of the function (split the function "
                   "or run it on one GPU)");
  const bool whole_local = sc.slab() && halo0 == 0 && inputs_fresh;
  const Box result_box = sc.local_box(result_decl);
  const Box bounds = sc.owned_bounds(bounds_decl, whole_local);
  neptune_hip_apply_geom_t g;
  fill_geom(g, result_box, bounds, in, NIN);
  int rc = geom_check_radius(&g, top_radius);
  if (rc == NEPTUNE_HIP_EOOB)
    die(sc.name(), "neptune_ir.apply reads outside an input's bounds (undefined behaviour in the reference lowering, "
                   "DataflowLowering.cpp:380-410); refusing to run it");
  if (rc != NEPTUNE_HIP_OK) die(sc.name(), "malformed neptune_ir.apply geometry");
  bool direct = dest != nullptr && dest->count == result_box.count();
  for (int k = 0; direct && k < NIN; ++k) direct = !overlaps(*dest, *in[k]);
  Val out;
  if (direct) {
    out = *dest;
    out.box = result_box;
  } else {
    out = sc.alloc(result_box, (int)sizeof(T));
  }
  out.stale_ghosts = sc.has_ghosts() && !whole_local;
  const void* ptrs[NIN];
  for (int k = 0; k < NIN; ++k) ptrs[k] = in[k]->dev;
  if (sc.pending()) {
    // The inputs' ghost planes are still being exchanged on another stream.  A stencil apply does its interior planes
    // now -- they read owned planes only -- then waits, then does the planes next to the ghosts and the ghost planes
    // themselves (copy-through); anything else waits first.
    const int64_t n0 = result_box.ub[0] - result_box.lb[0];
    const int64_t a = sc.ghost_lo() > 0 ? sc.ghost_lo() + halo0 : 0;
    const int64_t b = n0 - (sc.ghost_hi() > 0 ? sc.ghost_hi() + halo0 : 0);
    if (halo0 > 0 && !whole_local && b > a && !overlaps(out, *in[0])) {
      neptune_hip_apply_geom_t gi = g;
      gi.region_lb[0] = a;
      gi.region_ub[0] = b;
      rc = launch_apply<Body, T, RANK, NIN, FP>(body, &gi, ptrs, out.dev, sc.stream(), launch_override());
      if (rc != NEPTUNE_HIP_OK) die(sc.name(), "neptune_ir.apply launch rejected");
      sc.wait_pending();
      const int64_t edges[2][2] = {{0, a}, {b, n0}};
      for (auto& e : edges) {
        if (e[1] <= e[0]) continue;
        gi.region_lb[0] = e[0];
        gi.region_ub[0] = e[1];
        rc = launch_apply<Body, T, RANK, NIN, FP>(body, &gi, ptrs, out.dev, sc.stream(), launch_override());
        if (rc != NEPTUNE_HIP_OK) die(sc.name(), "neptune_ir.apply launch rejected");
      }
      if (direct) sc.mark_dirty(*dest);
      return out;
    }
    sc.wait_pending();
  }
  rc = launch_apply<Body, T, RANK, NIN, FP>(body, &g, ptrs, out.dev, sc.stream(), launch_override());
  if (rc != NEPTUNE_HIP_OK) die(sc.name(), "neptune_ir.apply launch rejected");
  if (direct) sc.mark_dirty(*dest);
  return out;
}

// One neptune_ir.apply of rank R = 4..6 whose accesses have no offset along the leading R-3 dimensions (batch / component
// dimensions; the reference's lowering is rank-generic, DataflowLowering.cpp:268-270, 301-308): for every leading index one
// rank-3 apply on the contiguous sub-field -- inside the leading bounds the body (which sees the leading indices as members
// `lead[]`), outside them the copy-through of input 0 (DataflowLowering.cpp:283-287).  Every input must cover the result's
// leading extent.  Slab mode: dim 0, the slab axis, is a leading dimension here, so the apply never reaches into a
// neighbouring plane: every rank runs its own leading indices (and the ghost planes too when its inputs' ghosts are good,
// like run_apply with halo0 == 0).
template <class Body, class T, int R, int NIN, class FP>
inline Val run_apply_batched(Scope& sc, Body body, const Box& result_global, const Box& bounds_decl, const Val* const* in,
                             const neptune_hip::Reach& top_radius, const Val* dest) {
  static_assert(R > 3 && R <= kMaxBoxRank, "run_apply_batched: rank 4..6");
  constexpr int L = R - 3;
  bool inputs_fresh = true;
  for (int k = 0; k < NIN; ++k) inputs_fresh = inputs_fresh && !in[k]->stale_ghosts;
  const bool whole_local = sc.slab() && inputs_fresh;
  const Box result_decl = sc.local_box(result_global);
  const Box bounds = sc.owned_bounds(bounds_decl, whole_local);
  auto sub_box = [](const Box& b) {
    Box r;
    r.rank = 3;
    for (int d = 0; d < 3; ++d) { r.lb[d] = b.lb[L + d]; r.ub[d] = b.ub[L + d]; }
    return r;
  };
  const Box out3 = sub_box(result_decl), bnd3 = sub_box(bounds);
  for (int k = 0; k < NIN; ++k) {
    if (in[k]->box.rank != R) die(sc.name(), "neptune_ir.apply: input rank differs from the result's");
    for (int d = 0; d < L; ++d)
      if (in[k]->box.lb[d] > result_decl.lb[d] || in[k]->box.ub[d] < result_decl.ub[d])
        die(sc.name(), "neptune_ir.apply of rank > 3: an input does not cover the result's leading extent");
  }
  bool direct = dest != nullptr && dest->count == result_decl.count();
  for (int k = 0; direct && k < NIN; ++k) direct = !overlaps(*dest, *in[k]);
  Val out;
  if (direct) {
    out = *dest;
    out.box = result_decl;
  } else {
    out = sc.alloc(result_decl, (int)sizeof(T));
  }
  out.stale_ghosts = sc.has_ghosts() && !whole_local;
  sc.wait_pending();
  const int64_t out_slab = out3.count();
  int64_t lead_n = 1;
  for (int d = 0; d < L; ++d) lead_n *= result_decl.ub[d] - result_decl.lb[d];
  Val sub[NIN];
  const Val* subp[NIN];
  const void* ptrs[NIN];
  for (int64_t flat = 0; flat < lead_n; ++flat) {
    int64_t idx[3] = {0, 0, 0}, rem = flat;
    for (int d = L - 1; d >= 0; --d) {
      const int64_t n = result_decl.ub[d] - result_decl.lb[d];
      idx[d] = result_decl.lb[d] + rem % n;
      rem /= n;
    }
    bool inside = true;
    for (int d = 0; d < L; ++d) inside = inside && idx[d] >= bounds.lb[d] && idx[d] < bounds.ub[d];
    for (int k = 0; k < NIN; ++k) {
      const Box& ib = in[k]->box;
      int64_t slab = 1, off = 0;
      for (int d = 0; d < 3; ++d) slab *= ib.ub[L + d] - ib.lb[L + d];
      for (int d = 0; d < L; ++d) off = off * (ib.ub[d] - ib.lb[d]) + (idx[d] - ib.lb[d]);
      sub[k] = *in[k];
      sub[k].box = sub_box(ib);
      sub[k].count = slab;
      sub[k].dev = static_cast<char*>(in[k]->dev) + off * slab * (int64_t)sizeof(T);
      subp[k] = &sub[k];
      ptrs[k] = sub[k].dev;
    }
    void* outp = static_cast<char*>(out.dev) + flat * out_slab * (int64_t)sizeof(T);
    neptune_hip_apply_geom_t g;
    Box b3 = bnd3;
    if (!inside) { for (int d = 0; d < 3; ++d) b3.ub[d] = b3.lb[d]; }   // zero trips: the kernel writes the copy-through only
    fill_geom(g, out3, b3, subp, NIN);
    int rc = inside ? geom_check_radius(&g, top_radius) : geom_validate(&g);
    if (rc == NEPTUNE_HIP_EOOB)
      die(sc.name(), "neptune_ir.apply reads outside an input's bounds (undefined behaviour in the reference lowering, "
                     "DataflowLowering.cpp:380-410); refusing to run it");
    if (rc != NEPTUNE_HIP_OK) die(sc.name(), "malformed neptune_ir.apply geometry");
    for (int d = 0; d < L; ++d) body.lead[d] = idx[d];
    rc = launch_apply<Body, T, 3, NIN, FP>(body, &g, ptrs, outp, sc.stream(), launch_override());
    if (rc != NEPTUNE_HIP_OK) die(sc.name(), "neptune_ir.apply launch rejected");
  }
  if (direct) sc.mark_dirty(*dest);
  return out;
}

// One neptune_ir.apply of rank R = 4..6 with offsets along its leading dimensions: a stencil in more than three dimensions
// (kernels/apply_nd.hpp).  Same contract as run_apply, slab mode included (dim 0 is the slab axis whatever the rank); the
// launch is not split around a pending halo exchange, it waits for it.
template <class Body, class T, int R, int NIN>
inline Val run_apply_nd(Scope& sc, const Body& body, const Box& result_decl, const Box& bounds_decl, const Val* const* in,
                        const neptune_hip::ReachN& top_reach, const Val* dest, int halo0 = 0) {
  static_assert(R > 3 && R <= kMaxBoxRank && R <= neptune_hip::kNdMaxRank, "run_apply_nd: rank 4..6");
  bool inputs_fresh = true;
  for (int k = 0; k < NIN; ++k) inputs_fresh = inputs_fresh && !in[k]->stale_ghosts;
  if (sc.has_ghosts() && halo0 > 0 && !inputs_fresh)
    die(sc.name(), "slab mode: neptune_ir.apply reads neighbouring planes of a value computed inside this call; "
                   "its ghost planes would need a halo exchange in the middle of the function (split the function "
                   "or run it on one GPU)");
  const bool whole_local = sc.slab() && halo0 == 0 && inputs_fresh;
  const Box result_box = sc.local_box(result_decl);
  const Box bounds = sc.owned_bounds(bounds_decl, whole_local);
  if (result_box.rank != R || bounds.rank != R) die(sc.name(), "malformed neptune_ir.apply geometry");
  for (int k = 0; k < NIN; ++k)
    if (in[k]->box.rank != R) die(sc.name(), "neptune_ir.apply: input rank differs from the result's");
  if (!in[0]->box.same_shape(result_box))
    die(sc.name(), "neptune_ir.apply: input 0 does not have the result's shape (DataflowLowering.cpp:283-287 copies it whole)");
  // every unconditional access of every cell inside apply.bounds (and the result) must stay inside its input's box
  for (int k = 0; k < NIN; ++k)
    for (int d = 0; d < R; ++d) {
      if (top_reach.hi[k][d] < top_reach.lo[k][d]) continue;
      const int64_t p0 = bounds.lb[d] > result_box.lb[d] ? bounds.lb[d] : result_box.lb[d];
      const int64_t p1 = bounds.ub[d] < result_box.ub[d] ? bounds.ub[d] : result_box.ub[d];
      bool empty = false;
      for (int e = 0; e < R; ++e) {
        const int64_t a = bounds.lb[e] > result_box.lb[e] ? bounds.lb[e] : result_box.lb[e];
        const int64_t b = bounds.ub[e] < result_box.ub[e] ? bounds.ub[e] : result_box.ub[e];
        empty = empty || b <= a;
      }
      if (empty) continue;
      if (p0 + top_reach.lo[k][d] < in[k]->box.lb[d] || p1 - 1 + top_reach.hi[k][d] >= in[k]->box.ub[d])
        die(sc.name(), "neptune_ir.apply reads outside an input's bounds (undefined behaviour in the reference lowering, "
                       "DataflowLowering.cpp:380-410); refusing to run it");
    }
  bool direct = dest != nullptr && dest->count == result_box.count();
  for (int k = 0; direct && k < NIN; ++k) direct = !overlaps(*dest, *in[k]);
  Val out;
  if (direct) {
    out = *dest;
    out.box = result_box;
  } else {
    out = sc.alloc(result_box, (int)sizeof(T));
  }
  out.stale_ghosts = sc.has_ghosts() && !whole_local;
  sc.wait_pending();
  neptune_hip::NdParams<T, NIN> P{};
  P.out = static_cast<T*>(out.dev);
  P.inner = 1;
  for (int d = 0; d < R; ++d) {
    P.n[d] = result_box.ub[d] - result_box.lb[d];
    P.olb[d] = result_box.lb[d];
    P.lb[d] = bounds.lb[d];
    P.ub[d] = bounds.ub[d];
    if (d > 0) P.inner *= P.n[d];
  }
  for (int k = 0; k < NIN; ++k) {
    P.in[k] = static_cast<const T*>(in[k]->dev);
    for (int d = 0; d < R; ++d) {
      P.m[k][d] = in[k]->box.ub[d] - in[k]->box.lb[d];
      P.sh[k][d] = result_box.lb[d] - in[k]->box.lb[d];
    }
  }
  P.r0 = 0;
  P.r1 = P.n[0];
  const int64_t total = (P.r1 - P.r0) * P.inner;
  if (total > 0) {
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) die(sc.name(), "neptune_ir.apply of rank > 3: the grid is not launchable");
    hipLaunchKernelGGL((neptune_hip::neptune_apply_nd<Body, T, R, NIN>), neptune_hip::grid_for_blocks(blocks), dim3(256), 0, sc.stream(), P,
                       body);
    NEPTUNE_HIP_CHECK(hipGetLastError());
    neptune_hip_note_launch(NEPTUNE_HIP_KERNEL_DIRECT, -1, 0);
  }
  if (direct) sc.mark_dirty(*dest);
  return out;
}

// rank 4..6 helpers: the leading R-3 dimensions of a box as a flat range of multi-indices, and the rank-3 sub-box / offset of
// the contiguous sub-buffer a leading multi-index names
inline Box last3(const Box& b) {
  Box r;
  r.rank = 3;
  const int L = b.rank - 3;
  for (int d = 0; d < 3; ++d) { r.lb[d] = b.lb[L + d]; r.ub[d] = b.ub[L + d]; }
  return r;
}
// cells from the start of `b`'s buffer to its sub-buffer at leading multi-index idx[0..L); -1 if idx lies outside b
inline int64_t lead_offset_cells(const Box& b, const int64_t* idx) {
  const int L = b.rank - 3;
  int64_t slab = 1, off = 0;
  for (int d = 0; d < 3; ++d) slab *= b.ub[L + d] - b.lb[L + d];
  for (int d = 0; d < L; ++d) {
    if (idx[d] < b.lb[d] || idx[d] >= b.ub[d]) return -1;
    off = off * (b.ub[d] - b.lb[d]) + (idx[d] - b.lb[d]);
  }
  return off * slab;
}
template <class F>
inline void for_each_lead(const Box& range, F&& f) {   // every leading multi-index of `range` (its first rank-3 dimensions), row-major
  const int L = range.rank - 3;
  int64_t n = 1;
  for (int d = 0; d < L; ++d) n *= range.ub[d] > range.lb[d] ? range.ub[d] - range.lb[d] : 0;
  for (int64_t flat = 0; flat < n; ++flat) {
    int64_t idx[3] = {0, 0, 0}, rem = flat;
    for (int d = L - 1; d >= 0; --d) {
      const int64_t e = range.ub[d] - range.lb[d];
      idx[d] = range.lb[d] + rem % e;
      rem /= e;
    }
    f(idx);
  }
}

// neptune_ir.store (DataflowLowering.cpp:165-220)
inline void run_store(Scope& sc, const Val& src, const Val& dst, const Box* bounds_decl, int dtype) {
  sc.wait_pending();   // a store may write planes the exchange is still sending
  int rc;
  Box clipped;
  const Box* bounds = bounds_decl;
  if (bounds_decl && sc.slab()) {  // the owned planes of the stored box
    clipped = sc.owned_bounds(*bounds_decl);
    bounds = &clipped;
    if (clipped.count() == 0) return;
  }
  if (!bounds) {
    if (src.dev == dst.dev) {  // the producing apply already wrote into the field
      sc.mark_dirty(dst);
      return;
    }
    rc = neptune_hip_store_full(dtype, src.dev, dst.dev, src.count, sc.stream());
  } else {
    if (src.box.rank > NEPTUNE_HIP_MAX_RANK) {
      // rank 4..6: one rank-3 box copy per leading index of the stored box
      const Box s3 = last3(src.box), d3 = last3(dst.box), b3 = last3(*bounds);
      rc = NEPTUNE_HIP_OK;
      for_each_lead(*bounds, [&](const int64_t* idx) {
        const int64_t so = lead_offset_cells(src.box, idx), doff = lead_offset_cells(dst.box, idx);
        if (so < 0 || doff < 0) { rc = NEPTUNE_HIP_EOOB; return; }
        const int r1 = neptune_hip_store_box(dtype, 3, static_cast<char*>(src.dev) + so * src.esize, s3.lb, s3.ub,
                                             static_cast<char*>(dst.dev) + doff * dst.esize, d3.lb, d3.ub, b3.lb, b3.ub, sc.stream());
        if (r1 != NEPTUNE_HIP_OK && rc == NEPTUNE_HIP_OK) rc = r1;
      });
    } else {
      rc = neptune_hip_store_box(dtype, src.box.rank, src.dev, src.box.lb, src.box.ub, dst.dev, dst.box.lb, dst.box.ub,
                                 bounds->lb, bounds->ub, sc.stream());
    }
  }
  if (rc == NEPTUNE_HIP_EOOB) die(sc.name(), "neptune_ir.store bounds leave a buffer");
  if (rc != NEPTUNE_HIP_OK) die(sc.name(), "neptune_ir.store rejected");
  sc.mark_dirty(dst);
}

// footprint / radius table of a two-input pointwise apply (time_advance's axpy step)
using PointwiseFP = Footprint<-1, 0, 0, 0, false, true>;
static const neptune_hip::Reach kPointwiseRadius2 = {{{0, 0, 0}, {0, 0, 0}, {1, 1, 1}, {1, 1, 1}},        // two inputs read at the centre,
                                                    {{0, 0, 0}, {0, 0, 0}, {-1, -1, -1}, {-1, -1, -1}}};  // the others not at all (hi < lo)

// The axpy step of an explicit time_advance on a field of rank 4..6: state + dt * k is pointwise over two dense buffers of
// one shape, so it runs as ONE rank-1 apply over the flat buffers (the rank-1 march kernel: the copy kernel's access
// pattern), whatever the rank -- and over everything this rank holds in slab mode, ghost planes included (they stay stale
// if either operand's are).
template <class T>
inline Val run_euler_axpy_flat(Scope& sc, T dt, const Val& state, const Val& k, const Val* dest) {
  if (!state.box.same_shape(k.box) || state.count != k.count)
    die(sc.name(), "neptune_ir.time_advance: rhs(state) does not have the state's shape");
  bool direct = dest != nullptr && dest->count == state.count && !overlaps(*dest, state) && !overlaps(*dest, k);
  Val out;
  if (direct) {
    out = *dest;
    out.box = state.box;
  } else {
    out = sc.alloc(state.box, (int)sizeof(T));
  }
  out.stale_ghosts = state.stale_ghosts || k.stale_ghosts;
  sc.wait_pending();
  if (state.count > 0) {
    neptune_hip_apply_geom_t g;
    memset(&g, 0, sizeof(g));
    g.rank = 1;
    g.num_inputs = 2;
    g.out_ub[0] = g.ub[0] = g.region_ub[0] = state.count;
    g.in_ub[0][0] = g.in_ub[1][0] = state.count;
    const void* ptrs[2] = {state.dev, k.dev};
    const int rc = launch_apply<neptune_hip::ops::EulerAxpy<T, 1>, T, 1, 2, PointwiseFP>(neptune_hip::ops::EulerAxpy<T, 1>{dt}, &g, ptrs, out.dev,
                                                                                         sc.stream(), launch_override());
    if (rc != NEPTUNE_HIP_OK) die(sc.name(), "neptune_ir.time_advance: axpy launch rejected");
  }
  if (direct) sc.mark_dirty(*dest);
  return out;
}

// neptune_ir.reduce {kind = "sum"} (DataflowLowering.cpp:589-698): blocking, result on the host
// slab mode: this rank's partial sum over its owned planes (the caller adds the ranks' results)
inline double run_reduce_sum(Scope& sc, const Val& src, const Box* bounds_decl, int dtype) {
  sc.wait_pending();
  double r = 0.0;
  Box clipped;
  const Box* bounds = bounds_decl;
  if (sc.slab() || bounds_decl) {
    clipped = sc.owned_bounds(bounds_decl ? *bounds_decl : src.box);
    bounds = &clipped;
    if (clipped.count() == 0) return 0.0;
  }
  if (src.box.rank > NEPTUNE_HIP_MAX_RANK) {
    // a whole-buffer sum of a field with leading batch dimensions: the same fixed tree over the flat buffer
    if (bounds) {
      // one rank-3 box sum per leading index of the reduced box, added up in index order
      const Box s3 = last3(src.box), b3 = last3(*bounds);
      for_each_lead(*bounds, [&](const int64_t* idx) {
        const int64_t so = lead_offset_cells(src.box, idx);
        if (so < 0) die(sc.name(), "neptune_ir.reduce bounds leave the input buffer");
        double part = 0.0;
        const int r1 = neptune_hip_reduce_sum(dtype, 3, static_cast<char*>(src.dev) + so * src.esize, s3.lb, s3.ub, b3.lb, b3.ub, &part,
                                              sc.stream());
        if (r1 == NEPTUNE_HIP_EOOB) die(sc.name(), "neptune_ir.reduce bounds leave the input buffer");
        if (r1 != NEPTUNE_HIP_OK) die(sc.name(), "neptune_ir.reduce rejected");
        r += part;
      });
      return r;
    }
    const int64_t flat_lb[1] = {0}, flat_ub[1] = {src.count};
    const int rc1 = neptune_hip_reduce_sum(dtype, 1, src.dev, flat_lb, flat_ub, nullptr, nullptr, &r, sc.stream());
    if (rc1 != NEPTUNE_HIP_OK) die(sc.name(), "neptune_ir.reduce rejected");
    return r;
  }
  const int rc = neptune_hip_reduce_sum(dtype, src.box.rank, src.dev, src.box.lb, src.box.ub, bounds ? bounds->lb : nullptr,
                                        bounds ? bounds->ub : nullptr, &r, sc.stream());
  if (rc == NEPTUNE_HIP_EOOB) die(sc.name(), "neptune_ir.reduce bounds leave the input buffer");
  if (rc != NEPTUNE_HIP_OK) die(sc.name(), "neptune_ir.reduce rejected");
  return r;
}

// neptune_ir.reduce {kind = "sum"} of a single-use neptune_ir.apply result, fused: the apply's values are
// summed as they are computed (csrc/kernels/reduce_apply.hpp); no intermediate temp, nothing written.
template <class Body, class T, int RANK, int NIN, class FP>
inline double run_apply_reduce_sum(Scope& sc, const Body& body, const Box& result_decl, const Box& bounds_decl,
                                   const Val* const* in,
                                   const neptune_hip::Reach& top_radius, int halo0,
                                   const Box* reduce_decl) {
  sc.wait_pending();
  for (int k = 0; k < NIN; ++k)
    if (sc.has_ghosts() && halo0 > 0 && in[k]->stale_ghosts)
      die(sc.name(), "slab mode: neptune_ir.apply reads neighbouring planes of a value computed inside this call");
  const Box result_box = sc.local_box(result_decl);
  const Box bounds = sc.owned_bounds(bounds_decl);
  const Box red = sc.owned_bounds(reduce_decl ? *reduce_decl : result_decl);
  neptune_hip_apply_geom_t g;
  fill_geom(g, result_box, bounds, in, NIN);
  int rc = geom_check_radius(&g, top_radius);
  if (rc == NEPTUNE_HIP_EOOB)
    die(sc.name(), "neptune_ir.apply reads outside an input's bounds (undefined behaviour in the reference lowering, "
                   "DataflowLowering.cpp:380-410); refusing to run it");
  if (rc != NEPTUNE_HIP_OK) die(sc.name(), "malformed neptune_ir.apply geometry");
  int64_t cells = 1;
  for (int d = 0; d < RANK; ++d) {
    const int64_t e = red.ub[d] - red.lb[d];
    if (e < 0) die(sc.name(), "malformed neptune_ir.reduce bounds");
    if (e > 0 && (red.lb[d] < result_box.lb[d] || red.ub[d] > result_box.ub[d]))
      die(sc.name(), "neptune_ir.reduce bounds leave the input buffer");
    g.region_lb[d] = red.lb[d] - result_box.lb[d];
    g.region_ub[d] = red.ub[d] - result_box.lb[d];
    cells *= e;
  }
  if (cells == 0) return 0.0;  // the reference's loop never runs, the accumulator stays 0
  const void* ptrs[NIN];
  for (int k = 0; k < NIN; ++k) ptrs[k] = in[k]->dev;
  DirectParams<T, NIN> P{};
  fill_direct_params<T, RANK, NIN>(&g, ptrs, nullptr, P);
  const int64_t eK = P.rub[2] - P.rlb[2];
  // the kernel keeps coordinates and row indices in 32 bits
  const int64_t lim = 0x7fffff00LL;
  bool narrow = P.n[0] * P.n[1] < lim && P.n[2] < lim;
  for (int k = 0; k < NIN; ++k) {
    narrow = narrow && P.m[k][0] * P.m[k][1] < lim && P.m[k][2] < lim;
    for (int ax = 0; ax < 3; ++ax) narrow = narrow && P.sh[k][ax] > -lim && P.sh[k][ax] < lim;
  }
  if (!narrow) die(sc.name(), "neptune_ir.reduce of an apply: fields with 2^31 rows or 2^31 cells per row are not supported");
  T* part = static_cast<T*>(neptune_hip_reduce_workspace());
  // pointwise body on 16-byte-aligned rows with all inputs in the result's box: the vector kernel
  constexpr int VK = 16 / (int)sizeof(T);
  bool vec = FP::MARCH_OK && FP::HALO_MASK == 0u && eK % VK == 0 && P.rlb[2] % VK == 0 && P.n[2] % VK == 0;
  for (int k = 0; k < NIN; ++k) {
    vec = vec && ((uintptr_t)ptrs[k] % 16 == 0);
    for (int ax = 0; ax < 3; ++ax) vec = vec && P.sh[k][ax] == 0 && P.m[k][ax] == P.n[ax];
  }
  const int cells_per_chunk = 256 * (vec ? VK : 1), iter = vec ? kReduceApplyIter / 2 : kReduceApplyIter;
  const int64_t nchunk = (eK + cells_per_chunk - 1) / cells_per_chunk;
  const int64_t trips = ((P.rub[0] - P.rlb[0]) * (P.rub[1] - P.rlb[1]) * nchunk + iter - 1) / iter;
  const int blocks = (int)(trips < kReduceBlocks ? trips : kReduceBlocks);
  if constexpr (FP::MARCH_OK && FP::HALO_MASK == 0u) {
    if (vec)
      hipLaunchKernelGGL((neptune_reduce_apply_vec<Body, T, RANK, NIN>), dim3(blocks), dim3(256), 0, sc.stream(), P, body, nchunk, part);
  } else {
    vec = false;
  }
  if (!vec)
    hipLaunchKernelGGL((neptune_reduce_apply<Body, T, RANK, NIN>), dim3(blocks), dim3(256), 0, sc.stream(), P, body, nchunk, part);
  hipLaunchKernelGGL(neptune_reduce_final<T>, dim3(1), dim3(256), 0, sc.stream(), part, blocks, part + kReduceBlocks);
  NEPTUNE_HIP_CHECK(hipGetLastError());
  T h = 0;
  NEPTUNE_HIP_CHECK(hipMemcpyAsync(&h, part + kReduceBlocks, sizeof(T), hipMemcpyDeviceToHost, sc.stream()));
  NEPTUNE_HIP_CHECK(hipStreamSynchronize(sc.stream()));
  return (double)h;
}

template <int RANK> struct MemRefOf;
template <> struct MemRefOf<1> { typedef NeptuneMemRef1D type; };
template <> struct MemRefOf<2> { typedef NeptuneMemRef2D type; };
template <> struct MemRefOf<3> { typedef NeptuneMemRef3D type; };
template <> struct MemRefOf<4> { typedef NeptuneMemRef4D type; };
template <> struct MemRefOf<5> { typedef NeptuneMemRef5D type; };
template <> struct MemRefOf<6> { typedef NeptuneMemRef6D type; };

// dense row-major descriptor over `p` (results: offset 0, stride[last] = 1; reference:
// NeptunePETScRuntime.cpp:881-892 view_x2D)
template <int RANK>
inline typename MemRefOf<RANK>::type make_memref(void* p, const Box& box) {
  typename MemRefOf<RANK>::type r;
  r.allocated = p;
  r.aligned = p;
  r.offset = 0;
  int64_t st = 1;
  for (int d = RANK - 1; d >= 0; --d) {
    r.sizes[d] = box.ub[d] - box.lb[d];
    r.strides[d] = st;
    st *= r.sizes[d];
  }
  return r;
}

}  // namespace lowered
}  // namespace neptune_hip

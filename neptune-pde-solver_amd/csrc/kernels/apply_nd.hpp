// apply_nd.hpp -- `neptune_ir.apply` of rank 4..6 whose accesses carry offsets along the LEADING dimensions too (a stencil
// in four or more dimensions; the reference's lowering is rank-generic: lib/Passes/DataflowLowering.cpp:268-270 builds one
// scf.for per dimension, :382-410 one index per dimension).  One lane per cell, every access a global load, 64-bit index
// arithmetic: the contract of neptune_apply_direct (apply_direct.hpp) with the rank as a template parameter -- a fresh
// result, physical copy-through of input 0 outside apply.bounds (:283-287), each input indexed through its own lower
// bounds, coordinates clamped into the input's buffer (only ever needed by cells whose value is discarded).
// Applies of rank 4..6 WITHOUT leading offsets do not come here: they are peeled into rank-3 launches of the fast kernels
// (lowered_runtime.hpp run_apply_batched).  Bound: L2 / HBM re-fetches like the direct kernel; no tuning, it exists so that
// no well-formed apply is refused.
#pragma once
#include "apply_common.hpp"

namespace neptune_hip {

constexpr int kNdMaxRank = 6;

// what the unconditional accesses of a body reach, per input and dimension (hi < lo: the input is never read unconditionally)
struct ReachN {
  int32_t lo[4][kNdMaxRank], hi[4][kNdMaxRank];
};

template <class T, int NIN>
struct NdParams {
  const T* in[NIN];
  T* out;
  int64_t n[kNdMaxRank];         // result (= input 0) physical extents
  int64_t olb[kNdMaxRank];       // result logical origin
  int64_t lb[kNdMaxRank], ub[kNdMaxRank];   // apply.bounds (logical)
  int64_t m[NIN][kNdMaxRank];    // input extents
  int64_t sh[NIN][kNdMaxRank];   // out_lb - in_lb
  int64_t r0, r1;                // planes (dim 0, result-physical) this launch writes
  int64_t inner;                 // cells per plane of dim 0
};

template <class T, int RANK, int NIN>
struct NdAcc {
  const NdParams<T, NIN>& P;
  int64_t q[RANK];  // result-physical coordinates of this lane's cell

  template <int IN, int... O>
  __device__ __forceinline__ T get() const {
    static_assert(IN >= 0 && IN < NIN, "input index out of range");
    static_assert(sizeof...(O) == RANK, "one offset per dimension");
    constexpr int off[RANK] = {O...};
    int64_t flat = 0;
#pragma unroll
    for (int d = 0; d < RANK; ++d) {
      int64_t c = q[d] + P.sh[IN][d] + off[d];
      c = c < 0 ? 0 : (c >= P.m[IN][d] ? P.m[IN][d] - 1 : c);
      flat = flat * P.m[IN][d] + c;
    }
    return P.in[IN][flat];
  }
  template <int D>
  __device__ __forceinline__ int64_t idx() const {
    static_assert(D >= 0 && D < RANK, "index argument out of range");
    return q[D] + P.olb[D];
  }
};

template <class Body, class T, int RANK, int NIN>
__global__ __launch_bounds__(256) void neptune_apply_nd(NdParams<T, NIN> P, Body body) {
  const int64_t total = (P.r1 - P.r0) * P.inner;
  const int64_t flat = linear_block() * blockDim.x + threadIdx.x;
  if (flat >= total) return;
  NdAcc<T, RANK, NIN> acc{P, {}};
  const int64_t o = P.r0 * P.inner + flat;     // dense row-major: the launch's planes are one contiguous run
  int64_t rem = o;
  bool inside = true;
#pragma unroll
  for (int d = RANK - 1; d >= 0; --d) {
    const int64_t c = d == 0 ? rem : rem % P.n[d];
    if (d != 0) rem /= P.n[d];
    acc.q[d] = c;
    const int64_t p = c + P.olb[d];
    inside = inside && p >= P.lb[d] && p < P.ub[d];
  }
  const T through = P.in[0][o];  // copy-through: physical-index-wise (DataflowLowering.cpp:283-287)
  const T val = body(acc);
  // (non-temporal: the result is not read again by this launch, and keeping it out of L2 leaves the neighbours' lines there --
  //  rows form 1024^3 7-point 2.52 -> 2.84 TB/s)
  __builtin_nontemporal_store(inside ? val : OutsideOf<Body, T>::apply(body, through), P.out + o);
}

}  // namespace neptune_hip

// apply_direct.hpp -- general `neptune_ir.apply` kernel: one lane per cell, every access a
// global load (neighbour reuse through L1/L2 only).
//
// Handles everything the reference lowering does (lib/Passes/DataflowLowering.cpp:258-448):
// rank 1..3, several inputs each with its own logical box, any offsets, odd sizes,
// sub-regions.  It is the correctness workhorse and the fallback for shapes the march
// kernel (apply_march.hpp) declines; pointwise applies already stream at full rate here.
#pragma once
#include "apply_common.hpp"

namespace neptune_hip {

// All per-axis arrays are in kernel axis order (I, J, K); absent axes have extent 1.
template <class T, int NIN>
struct DirectParams {
  const T* in[NIN];
  T* out;
  int64_t n[3];        // result (= input 0) physical extents
  int64_t olb[3];      // result logical origin
  int64_t lb[3], ub[3];  // apply.bounds (logical)
  int64_t m[NIN][3];   // input extents
  int64_t sh[NIN][3];  // out_lb - in_lb : result-physical -> input-physical shift
  int64_t rlb[3], rub[3];  // region of this launch (result-physical)
};

template <class T, int RANK, int NIN>
struct DirectAcc {
  const DirectParams<T, NIN>& P;
  int64_t q[3];  // result-physical coordinates of this lane's cell

  // access %in_IN[O...] : physical = logical + off - in_lb  (DataflowLowering.cpp:380-410).
  // Coordinates are clamped into the input's buffer: for a cell inside apply.bounds the plan
  // check guarantees they already are; for a copy-through cell the value is discarded.
  template <int IN, int... O>
  __device__ __forceinline__ T get() const {
    static_assert(IN >= 0 && IN < NIN, "input index out of range");
    constexpr int oi = PickOffset<RANK, AxisMap<RANK>::I, O...>::value;
    constexpr int oj = PickOffset<RANK, AxisMap<RANK>::J, O...>::value;
    constexpr int ok = PickOffset<RANK, AxisMap<RANK>::K, O...>::value;
    int64_t ci = q[0] + P.sh[IN][0] + oi;
    int64_t cj = q[1] + P.sh[IN][1] + oj;
    int64_t ck = q[2] + P.sh[IN][2] + ok;
    ci = ci < 0 ? 0 : (ci >= P.m[IN][0] ? P.m[IN][0] - 1 : ci);
    cj = cj < 0 ? 0 : (cj >= P.m[IN][1] ? P.m[IN][1] - 1 : cj);
    ck = ck < 0 ? 0 : (ck >= P.m[IN][2] ? P.m[IN][2] - 1 : ck);
    return P.in[IN][(ci * P.m[IN][1] + cj) * P.m[IN][2] + ck];
  }
  // region index argument #D (logical coordinate of the current point)
  template <int D>
  __device__ __forceinline__ int64_t idx() const {
    static_assert(D >= 0 && D < RANK, "index argument out of range");
    constexpr int ax = (RANK == 3) ? D : (RANK == 2 ? (D == 0 ? 0 : 2) : 2);
    return q[ax] + P.olb[ax];
  }
};

template <class Body, class T, int RANK, int NIN>
__global__ __launch_bounds__(256) void neptune_apply_direct(DirectParams<T, NIN> P, Body body) {
  const int64_t eK = P.rub[2] - P.rlb[2], eJ = P.rub[1] - P.rlb[1], eI = P.rub[0] - P.rlb[0];
  const int64_t total = eI * eJ * eK;
  const int64_t flat = linear_block() * blockDim.x + threadIdx.x;
  if (flat >= total) return;
  DirectAcc<T, RANK, NIN> acc{P, {0, 0, 0}};
  const int64_t row = flat / eK;
  acc.q[2] = P.rlb[2] + (flat - row * eK);
  acc.q[1] = P.rlb[1] + row % eJ;
  acc.q[0] = P.rlb[0] + row / eJ;

  const int64_t o = (acc.q[0] * P.n[1] + acc.q[1]) * P.n[2] + acc.q[2];
  bool inside = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int64_t p = acc.q[a] + P.olb[a];
    inside = inside && p >= P.lb[a] && p < P.ub[a];
  }
  const T through = P.in[0][o];  // copy-through: physical-index-wise (DataflowLowering.cpp:283-287)
  const T val = body(acc);
  // (non-temporal: the result is not read again by this launch, and keeping it out of L2 leaves the neighbours' lines there --
  //  rows form 1024^3 7-point 2.52 -> 2.84 TB/s)
  __builtin_nontemporal_store(inside ? val : OutsideOf<Body, T>::apply(body, through), P.out + o);
}

// wave-uniform pointer -> SGPR pair, so the load takes the "scalar base + 32-bit lane offset" form
template <class T>
__device__ __forceinline__ const T* uniform_ptr(const T* p) {
  const uint64_t u = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return reinterpret_cast<const T*>(((uint64_t)hi << 32) | lo);
}

// Accessor of the row-wise kernels (neptune_apply_rows below, neptune_reduce_apply): a wave works on ONE row at a time, so the I and J coordinates (and with
// them each input row's base address, clamps included) are wave-uniform scalar work; per lane there is
// a 32-bit K coordinate, one add and two clamps per access.  Same contract as DirectAcc: coordinates
// are clamped into the input's buffer, the plan check guarantees in-bounds cells never need it.
// Host side guarantees that every extent, every origin shift and every buffer's row count fit 31 bits.
template <class T, int RANK, int NIN>
struct RowAcc {
  const DirectParams<T, NIN>& P;
  int32_t qi, qj;  // result-physical row coordinates (uniform)
  int32_t qk;      // result-physical K coordinate of this lane's cell

  // first element of row (ci, cj) of a buffer with extents m[]: 32-bit row index, one widening multiply
  static __device__ __forceinline__ const T* row_ptr(const T* base, int32_t ci, int32_t cj, const int64_t (&m)[3]) {
    const uint32_t r = (uint32_t)ci * (uint32_t)m[1] + (uint32_t)cj;
    return uniform_ptr(base + (uint64_t)r * (uint32_t)m[2]);
  }
  template <int IN, int... O>
  __device__ __forceinline__ T get() const {
    static_assert(IN >= 0 && IN < NIN, "input index out of range");
    constexpr int oi = PickOffset<RANK, AxisMap<RANK>::I, O...>::value;
    constexpr int oj = PickOffset<RANK, AxisMap<RANK>::J, O...>::value;
    constexpr int ok = PickOffset<RANK, AxisMap<RANK>::K, O...>::value;
    int32_t ci = qi + (int32_t)P.sh[IN][0] + oi;
    int32_t cj = qj + (int32_t)P.sh[IN][1] + oj;
    int32_t ck = qk + (int32_t)P.sh[IN][2] + ok;
    const int32_t li = (int32_t)P.m[IN][0] - 1, lj = (int32_t)P.m[IN][1] - 1, lk = (int32_t)P.m[IN][2] - 1;
    ci = ci < 0 ? 0 : (ci > li ? li : ci);
    cj = cj < 0 ? 0 : (cj > lj ? lj : cj);
    ck = ck < 0 ? 0 : (ck > lk ? lk : ck);
    return row_ptr(P.in[IN], ci, cj, P.m[IN])[(uint32_t)ck];
  }
  template <int D>
  __device__ __forceinline__ int64_t idx() const {
    static_assert(D >= 0 && D < RANK, "index argument out of range");
    constexpr int ax = (RANK == 3) ? D : (RANK == 2 ? (D == 0 ? 0 : 2) : 2);
    return (int64_t)(ax == 0 ? qi : (ax == 1 ? qj : qk)) + P.olb[ax];
  }
};

// ---- rows form of the direct kernel ---------------------------------------------------------------------
// Same contract as neptune_apply_direct, different work assignment: a workgroup owns ONE 256-cell chunk
// of one row, so everything but the K coordinate is wave-uniform: row base addresses (clamps and the
// multiplies included) are scalar work done once per access, the per-lane part of an access is an add,
// two clamps and a load with a scalar base.  The flat form above spends ~190 vector instructions per
// cell on 64-bit index arithmetic; this one a few (27-point fp32: 1.8 vs 0.9 TB/s; 7-point: +3..5 %,
// it is bound by neighbour re-fetches, not instructions).  Workgroups keep the hardware's round-robin
// order over the XCDs: each XCD then holds a 1/8 share of three consecutive planes in its L2
// (measured better than contiguous runs per XCD, which push the plane-to-plane reuse out of the L2).
// Needs every extent, shift and row count to fit 31 bits (host-checked); otherwise the flat form runs.
template <class Body, class T, int RANK, int NIN>
__global__ __launch_bounds__(256) void neptune_apply_rows(DirectParams<T, NIN> P, Body body, uint32_t nchunk) {
  const int64_t b = linear_block();
  const int32_t eJ = (int32_t)(P.rub[1] - P.rlb[1]), eK = (int32_t)(P.rub[2] - P.rlb[2]);
  if (b >= (P.rub[0] - P.rlb[0]) * (int64_t)eJ * nchunk) return;  // the folded grid's last row of workgroups
  const uint32_t row = (uint32_t)(b / nchunk), c = (uint32_t)(b - (int64_t)row * nchunk);
  const int32_t i = (int32_t)(row / (uint32_t)eJ), j = (int32_t)(row - (uint32_t)i * (uint32_t)eJ);
  const int32_t k = (int32_t)c * 256 + (int32_t)threadIdx.x;
  if (k >= eK) return;
  RowAcc<T, RANK, NIN> a{P, (int32_t)P.rlb[0] + i, (int32_t)P.rlb[1] + j, (int32_t)P.rlb[2] + k};
  const int64_t pi = a.qi + P.olb[0], pj = a.qj + P.olb[1], pk = a.qk + P.olb[2];
  const bool inside = pi >= P.lb[0] && pi < P.ub[0] && pj >= P.lb[1] && pj < P.ub[1] && pk >= P.lb[2] && pk < P.ub[2];
  const T* in0 = RowAcc<T, RANK, NIN>::row_ptr(P.in[0], a.qi, a.qj, P.n);
  T* out = const_cast<T*>(RowAcc<T, RANK, NIN>::row_ptr(P.out, a.qi, a.qj, P.n));
  const T through = in0[(uint32_t)a.qk];  // copy-through: physical-index-wise (DataflowLowering.cpp:283-287)
  const T val = body(a);
  __builtin_nontemporal_store(inside ? val : OutsideOf<Body, T>::apply(body, through), out + (uint32_t)a.qk);
}

}  // namespace neptune_hip

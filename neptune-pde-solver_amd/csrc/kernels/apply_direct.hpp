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
  const int64_t flat = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
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
  P.out[o] = inside ? val : OutsideOf<Body, T>::apply(body, through);
}

}  // namespace neptune_hip

// reduce_apply.hpp -- `neptune_ir.reduce {kind = "sum"}` of a single-use `neptune_ir.apply` result in
// ONE pass: the apply's value at every cell of the reduced box is computed in registers and summed,
// the intermediate temp never exists in memory.  This is the shape of the dot products and norms a
// Krylov iteration needs (reduce(apply(u*v)), reduce(apply(u*u))): the inputs are read once, nothing
// is written -- 2 field passes for a dot product instead of the 4 of apply-then-reduce.
//
// Semantics are those of the two ops back to back (DataflowLowering.cpp:258-448, 589-698): a cell
// inside apply.bounds contributes body(p), a cell outside contributes input 0 (copy-through).
// Summation order: a FIXED tree like neptune_reduce_* (util_kernels.hpp) -- workgroups own contiguous
// runs of 256-cell row chunks, lanes take consecutive cells (coalesced), per-lane serial sum, wave
// shuffle tree, LDS, then neptune_reduce_final adds the per-workgroup partials in index order.  Reproducible
// run to run; differs from the reference's serial sum by rounding within 2(n-1) eps sum|x_i|.
#pragma once
#include "apply_direct.hpp"
#include "util_kernels.hpp"

namespace neptune_hip {

constexpr int kReduceApplyIter = 8;  // row chunks per lane per trip, all their loads in flight together

// A *row chunk* is 256 consecutive cells of one row of the reduced box (one cell per lane, coalesced).
// Workgroups own contiguous runs of row chunks; the (row, chunk) -> (i, j, k) bookkeeping is
// workgroup-uniform, i.e. scalar work.  P.rlb / P.rub hold the REDUCED box in result-physical
// coordinates; P.out is unused.
template <class Body, class T, int RANK, int NIN>
__global__ __launch_bounds__(256) void neptune_reduce_apply(DirectParams<T, NIN> P, Body body, int64_t nchunk,
                                                             T* __restrict__ partials) {
  __shared__ T lds[4];
  const int32_t eJ = (int32_t)(P.rub[1] - P.rlb[1]), eK = (int32_t)(P.rub[2] - P.rlb[2]);
  const int32_t r0 = (int32_t)P.rlb[0], r1 = (int32_t)P.rlb[1], r2 = (int32_t)P.rlb[2];
  const int64_t total = (P.rub[0] - P.rlb[0]) * eJ * nchunk;
  const int64_t per = (total + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per;
  const int64_t hi = lo + per < total ? lo + per : total;
  T acc = 0;
  for (int64_t rc0 = lo; rc0 < hi; rc0 += kReduceApplyIter) {
    const int64_t row0 = rc0 / nchunk;
    const int32_t c0 = (int32_t)(rc0 - row0 * nchunk), i0 = (int32_t)(row0 / eJ), j0 = (int32_t)(row0 - (int64_t)i0 * eJ);
    int32_t i = i0, j = j0, c = c0;
    T v[kReduceApplyIter];
#pragma unroll
    for (int it = 0; it < kReduceApplyIter; ++it) {
      const bool live = rc0 + it < hi;  // uniform; a dead slot re-reads the first chunk and is discarded
      RowAcc<T, RANK, NIN> a{P, r0 + (live ? i : i0), r1 + (live ? j : j0), 0};
      const int32_t k = (live ? c : c0) * 256 + (int32_t)threadIdx.x;
      const bool valid = live && k < eK;
      a.qk = r2 + (k < eK ? k : eK - 1);  // clamped: the loads stay inside the buffers
      const int64_t pi = a.qi + P.olb[0], pj = a.qj + P.olb[1], pk = a.qk + P.olb[2];
      const bool inside = pi >= P.lb[0] && pi < P.ub[0] && pj >= P.lb[1] && pj < P.ub[1] && pk >= P.lb[2] && pk < P.ub[2];
      const T through = RowAcc<T, RANK, NIN>::row_ptr(P.in[0], a.qi, a.qj, P.n)[(uint32_t)a.qk];
      const T val = body(a);
      v[it] = valid ? (inside ? val : OutsideOf<Body, T>::apply(body, through)) : (T)0;
      if (++c == (int32_t)nchunk) {
        c = 0;
        if (++j == eJ) { j = 0; ++i; }
      }
    }
#pragma unroll
    for (int it = 0; it < kReduceApplyIter; ++it) acc += v[it];
  }
  const T r = block_sum(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = r;
}


// ---- pointwise bodies on aligned rows: 16-byte loads ----------------------------------------------------
// When every access of the body is at offset 0 (dot products, norms, weighted sums) and rows are 16-byte
// aligned multiples of VK cells with all inputs sharing the result's box, a lane takes VK adjacent cells
// with ONE 16-byte load per input; the body is evaluated per element on the loaded vectors.  A row chunk
// is then 256*VK cells.  Same summation tree shape (per-lane serial over its cells in index order).
template <class T, int RANK, int NIN>
struct PointVecAcc {
  static constexpr int VK = 16 / sizeof(T);
  typedef T vec __attribute__((ext_vector_type(VK)));
  const DirectParams<T, NIN>& P;
  int32_t qi, qj, qk;  // result-physical coordinates of element 0
  vec x[NIN];
  int e;               // element under evaluation (compile-time after unrolling)
  template <int IN, int... O>
  __device__ __forceinline__ T get() const {
    static_assert(IN >= 0 && IN < NIN, "input index out of range");
    static_assert(((O == 0) && ...), "pointwise body expected");
    return x[IN][e];
  }
  template <int D>
  __device__ __forceinline__ int64_t idx() const {
    static_assert(D >= 0 && D < RANK, "index argument out of range");
    constexpr int ax = (RANK == 3) ? D : (RANK == 2 ? (D == 0 ? 0 : 2) : 2);
    return (int64_t)(ax == 0 ? qi : (ax == 1 ? qj : qk + e)) + P.olb[ax];
  }
};

template <class Body, class T, int RANK, int NIN>
__global__ __launch_bounds__(256) void neptune_reduce_apply_vec(DirectParams<T, NIN> P, Body body, int64_t nchunk,
                                                                 T* __restrict__ partials) {
  constexpr int VK = 16 / sizeof(T);
  constexpr int ITER = kReduceApplyIter / 2;
  static_assert(ITER == 4, "the accessor array below is spelled out for 4 slots");
  typedef typename PointVecAcc<T, RANK, NIN>::vec vec;
  __shared__ T lds[4];
  const int32_t eJ = (int32_t)(P.rub[1] - P.rlb[1]), eK = (int32_t)(P.rub[2] - P.rlb[2]);
  const int32_t r0 = (int32_t)P.rlb[0], r1 = (int32_t)P.rlb[1], r2 = (int32_t)P.rlb[2];
  const int64_t total = (P.rub[0] - P.rlb[0]) * eJ * nchunk;
  const int64_t per = (total + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per;
  const int64_t hi = lo + per < total ? lo + per : total;
  T acc = 0;
  for (int64_t rc0 = lo; rc0 < hi; rc0 += ITER) {
    const int64_t row0 = rc0 / nchunk;
    const int32_t c0 = (int32_t)(rc0 - row0 * nchunk), i0 = (int32_t)(row0 / eJ), j0 = (int32_t)(row0 - (int64_t)i0 * eJ);
    int32_t i = i0, j = j0, c = c0;
    PointVecAcc<T, RANK, NIN> a[ITER] = {{P, 0, 0, 0, {}, 0}, {P, 0, 0, 0, {}, 0}, {P, 0, 0, 0, {}, 0}, {P, 0, 0, 0, {}, 0}};
    bool valid[ITER], inside_ij[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const bool live = rc0 + it < hi;
      a[it].qi = r0 + (live ? i : i0);
      a[it].qj = r1 + (live ? j : j0);
      const int32_t k = ((live ? c : c0) * 256 + (int32_t)threadIdx.x) * VK;
      valid[it] = live && k < eK;               // eK is a multiple of VK: a vector is all in or all out
      a[it].qk = r2 + (k < eK ? k : eK - VK);
      const int64_t pi = a[it].qi + P.olb[0], pj = a[it].qj + P.olb[1];
      inside_ij[it] = pi >= P.lb[0] && pi < P.ub[0] && pj >= P.lb[1] && pj < P.ub[1];
      static_for<NIN>([&](auto nc) {
        constexpr int n = nc;  // all inputs share the result's box (host-checked): same row, same k
        const T* row = RowAcc<T, RANK, NIN>::row_ptr(P.in[n], a[it].qi, a[it].qj, P.n);
        a[it].x[n] = *reinterpret_cast<const vec*>(row + (uint32_t)a[it].qk);
      });
      if (++c == (int32_t)nchunk) {
        c = 0;
        if (++j == eJ) { j = 0; ++i; }
      }
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
      for (int e = 0; e < VK; ++e) {
        a[it].e = e;
        const int64_t pk = (int64_t)a[it].qk + e + P.olb[2];
        const bool inside = inside_ij[it] && pk >= P.lb[2] && pk < P.ub[2];
        const T val = body(a[it]);
        const T v = valid[it] ? (inside ? val : OutsideOf<Body, T>::apply(body, a[it].x[0][e])) : (T)0;
        acc += v;
      }
    }
  }
  const T r = block_sum(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

}  // namespace neptune_hip

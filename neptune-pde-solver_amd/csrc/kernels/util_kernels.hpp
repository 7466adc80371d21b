// util_kernels.hpp -- store (box copy), streaming copy, deterministic fill and bitwise
// compare kernels around the apply kernels.
#pragma once
#include "apply_common.hpp"

namespace neptune_hip {

// neptune_ir.store with bounds (lib/Passes/DataflowLowering.cpp:184-217): the logical box
// [lb,ub) is copied between two buffers that each have their own logical origin.
// Per-axis arrays in (I,J,K) order; absent axes extent 1.
struct BoxCopyParams {
  int64_t ext[3];     // ub - lb
  int64_t soff[3];    // lb - src_lb
  int64_t doff[3];    // lb - dst_lb
  int64_t sshape[3];  // src buffer extents
  int64_t dshape[3];  // dst buffer extents
};

template <class T>
__global__ __launch_bounds__(256) void neptune_store_box(const T* __restrict__ src, T* __restrict__ dst,
                                                         BoxCopyParams P) {
  // one workgroup = one chunk of 256*VK consecutive cells of one row of the box: the row decode is
  // workgroup-uniform scalar work, a lane moves VK adjacent cells with one 16-byte load and store (rows of a
  // sub-box start anywhere in either buffer: unaligned accesses); the row's last, partial vector goes cell by cell
  constexpr int VK = 16 / sizeof(T);
  typedef T uvec __attribute__((ext_vector_type(VK), aligned(sizeof(T))));
  const int64_t nchunk = (P.ext[2] + 256 * VK - 1) / (256 * VK);
  const int64_t b = linear_block();
  if (b >= P.ext[0] * P.ext[1] * nchunk) return;
  const int64_t row = b / nchunk, c = b - row * nchunk;
  const int64_t i = row / P.ext[1], j = row - i * P.ext[1];
  const int64_t k0 = (c * 256 + threadIdx.x) * VK;
  if (k0 >= P.ext[2]) return;
  const T* s = src + ((i + P.soff[0]) * P.sshape[1] + (j + P.soff[1])) * P.sshape[2] + (k0 + P.soff[2]);
  T* d = dst + ((i + P.doff[0]) * P.dshape[1] + (j + P.doff[1])) * P.dshape[2] + (k0 + P.doff[2]);
  if (k0 + VK <= P.ext[2]) {
    __builtin_nontemporal_store(*reinterpret_cast<const uvec*>(s), reinterpret_cast<uvec*>(d));
  } else {
    for (int e = 0; k0 + e < P.ext[2]; ++e) d[e] = s[e];
  }
}

// 16 B per lane streaming copies: the measured HBM ceiling the apply kernels are compared with.
// n16 = number of 16-byte words.
//   grid-stride form (mode 0)
__global__ __launch_bounds__(256) void neptune_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                      int64_t n16) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}
//   U loads in flight per lane, exact grid, optional non-temporal loads/stores (modes 1..)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int U, bool NTLD, bool NTST>
__global__ __launch_bounds__(256) void neptune_copy16_unrolled(const u32x4* __restrict__ src,
                                                               u32x4* __restrict__ dst, int64_t n16) {
  const int64_t base = (int64_t)blockIdx.x * (256 * U) + threadIdx.x;
  u32x4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = base + (int64_t)u * 256;
    if (i < n16) v[u] = NTLD ? __builtin_nontemporal_load(src + i) : src[i];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = base + (int64_t)u * 256;
    if (i < n16) {
      if (NTST) __builtin_nontemporal_store(v[u], dst + i);
      else dst[i] = v[u];
    }
  }
}

// Krylov vector updates on device-resident vectors (the host KSP loop of the reference updates its
// Vecs on the host, NeptunePETScRuntime.cpp:719-786).  NeptuneIR regions are IsolatedFromAbove, so a run-time scalar
// such as CG's alpha cannot enter an apply; these two kernels are the missing piece for a solver loop that never leaves
// the GPU.  Two roundings each (the product, then the sum), no FMA -- what numpy's `y += a * x` computes.
//   AXPY: y[i] = y[i] + a * x[i]        XPAY: y[i] = x[i] + a * y[i]
template <class T, bool XPAY>
__global__ __launch_bounds__(256) void neptune_vec_update(int64_t n, T a, const T* __restrict__ x, T* __restrict__ y) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if constexpr (XPAY) { const T t = a * y[i]; y[i] = x[i] + t; }
    else { const T t = a * x[i]; y[i] = y[i] + t; }
  }
}

// the same update on 16-byte-aligned vectors: one 16-byte load of x and of y per lane, exact grid, non-temporal store (the
// access pattern of the fastest copy kernel: grid-stride loops of scalar accesses run ~25 % below it); the n % VK elements
// at the end go through lane 0 of the first workgroup
template <class T, bool XPAY>
__global__ __launch_bounds__(256) void neptune_vec_update_v(int64_t n, T a, const T* __restrict__ x, T* __restrict__ y) {
  constexpr int VK = 16 / (int)sizeof(T);
  typedef T V __attribute__((ext_vector_type(VK)));
  const int64_t nv = n / VK;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < nv) {
    const V xv = reinterpret_cast<const V*>(x)[i];
    const V yv = reinterpret_cast<const V*>(y)[i];
    V r;
#pragma unroll
    for (int e = 0; e < VK; ++e) {
      if constexpr (XPAY) { const T t = a * yv[e]; r[e] = xv[e] + t; }
      else { const T t = a * xv[e]; r[e] = yv[e] + t; }
    }
    __builtin_nontemporal_store(r, reinterpret_cast<V*>(y) + i);
  }
  if (i == 0)
    for (int64_t j = nv * VK; j < n; ++j) {
      if constexpr (XPAY) { const T t = a * y[j]; y[j] = x[j] + t; }
      else { const T t = a * x[j]; y[j] = y[j] + t; }
    }
}

template <class T>
__global__ __launch_bounds__(256) void neptune_fill_hash(T* __restrict__ dst, int64_t count, int64_t index_offset,
                                                         uint64_t seed) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
    if constexpr (sizeof(T) == 8) dst[i] = hash_f64(i + index_offset, seed);
    else dst[i] = hash_f32(i + index_offset, seed);
  }
}

// counts elements whose bit patterns differ (NaN-safe, -0.0 != +0.0)
template <class U>
__global__ __launch_bounds__(256) void neptune_count_mismatch(const U* __restrict__ a, const U* __restrict__ b,
                                                              int64_t count, unsigned long long* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long local = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) local += (a[i] != b[i]);
  // wave reduction, then one atomic per wave
  for (int off = kWave / 2; off > 0; off >>= 1) local += __shfl_down(local, off);
  if ((threadIdx.x & (kWave - 1)) == 0 && local) atomicAdd(out, local);
}

// ---- neptune_ir.reduce {kind = "sum"} (lib/Passes/DataflowLowering.cpp:589-698) ------------------
// The reference sums serially in row-major order.  A GPU cannot keep that order; this reduction
// uses a FIXED tree instead (lane-strided partial sums -> wave shuffle tree -> LDS -> one partial
// per workgroup -> second kernel adds the partials in index order), so the result is bit-for-bit
// reproducible from run to run and independent of scheduling (no atomics), but differs from the
// serial sum by rounding: |gpu - serial| <= 2 (n-1) eps sum|x_i| (each order is within (n-1) eps
// sum|x_i| of the exact sum).  Accumulation is in the element type, like the reference.
constexpr int kReduceBlocks = 2048;  // partials of the first pass; the workspace holds kReduceBlocks + 1 elements

struct ReduceBoxParams {
  int64_t ext[3];     // reduced box extents (I,J,K order, absent axes 1)
  int64_t off[3];     // box origin - buffer origin
  int64_t shape[3];   // buffer extents
};

template <class T>
__device__ __forceinline__ T block_sum(T v, T* lds /* >= blockDim.x / 64 entries */) {
  for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
  if (lane == 0) lds[w] = v;
  __syncthreads();
  T r = 0;
  if (threadIdx.x == 0) {
    const int nw = blockDim.x / kWave;
    for (int i = 0; i < nw; ++i) r += lds[i];  // fixed order
  }
  return r;  // valid in thread 0
}

// contiguous buffer: every workgroup owns one contiguous slice, lanes stride through it with
// 16-byte loads (VK cells per lane per load, VK independent partial sums per lane)
template <class T>
__global__ __launch_bounds__(256) void neptune_reduce_partial_flat(const T* __restrict__ src, int64_t count,
                                                                    T* __restrict__ partials) {
  constexpr int VK = 16 / sizeof(T);
  typedef T vec __attribute__((ext_vector_type(VK)));
  __shared__ T lds[4];
  // slices are whole numbers of 16-byte words; the last workgroup also takes the scalar tail
  const int64_t nvec = count / VK;
  const int64_t per = (nvec + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per;
  const int64_t hi = lo + per < nvec ? lo + per : nvec;
  const vec* __restrict__ sv = reinterpret_cast<const vec*>(src);
  T part[VK];
#pragma unroll
  for (int e = 0; e < VK; ++e) part[e] = 0;
#pragma unroll 4
  for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const vec x = sv[i];
#pragma unroll
    for (int e = 0; e < VK; ++e) part[e] += x[e];
  }
  T acc = 0;
#pragma unroll
  for (int e = 0; e < VK; ++e) acc += part[e];
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
    for (int64_t i = nvec * VK; i < count; ++i) acc += src[i];
  const T r = block_sum(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

// sub-box of a buffer (e.g. the interior of a field): workgroups own contiguous runs of row chunks -- a chunk is
// 256*VK consecutive cells of one row of the box, VK adjacent cells per lane in one 16-byte load (rows of a sub-box
// start anywhere: unaligned loads) -- so the (row, chunk) -> (i, j, k) bookkeeping is workgroup-uniform scalar work
// instead of a 64-bit division per cell.  Four chunks per trip keep their loads in flight together.
template <class T>
__global__ __launch_bounds__(256) void neptune_reduce_partial_box(const T* __restrict__ src, ReduceBoxParams P,
                                                                   T* __restrict__ partials) {
  constexpr int VK = 16 / sizeof(T), ITER = 4;
  typedef T uvec __attribute__((ext_vector_type(VK), aligned(sizeof(T))));
  __shared__ T lds[4];
  const int64_t cells_per_chunk = 256 * VK;
  const int64_t nchunk = (P.ext[2] + cells_per_chunk - 1) / cells_per_chunk;
  const int64_t total = P.ext[0] * P.ext[1] * nchunk;
  const int64_t per = (total + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per;
  const int64_t hi = lo + per < total ? lo + per : total;
  T part[VK];
#pragma unroll
  for (int e = 0; e < VK; ++e) part[e] = 0;
  for (int64_t rc0 = lo; rc0 < hi; rc0 += ITER) {
    const int64_t row0 = rc0 / nchunk;
    int64_t c = rc0 - row0 * nchunk, i = row0 / P.ext[1], j = row0 - i * P.ext[1];
    T v[ITER][VK];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const bool live = rc0 + it < hi;  // uniform
      const int64_t k0 = (c * 256 + threadIdx.x) * VK;
      const T* p = src + ((i + P.off[0]) * P.shape[1] + (j + P.off[1])) * P.shape[2] + (P.off[2] + k0);
      if (live && k0 + VK <= P.ext[2]) {
        const uvec x = *reinterpret_cast<const uvec*>(p);
#pragma unroll
        for (int e = 0; e < VK; ++e) v[it][e] = x[e];
      } else {
#pragma unroll
        for (int e = 0; e < VK; ++e) v[it][e] = (live && k0 + e < P.ext[2]) ? p[e] : (T)0;  // the row's last, partial vector
      }
      if (live && ++c == nchunk) {
        c = 0;
        if (++j == P.ext[1]) { j = 0; ++i; }
      }
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it)
#pragma unroll
      for (int e = 0; e < VK; ++e) part[e] += v[it][e];
  }
  T acc = 0;
#pragma unroll
  for (int e = 0; e < VK; ++e) acc += part[e];
  const T r = block_sum(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

template <class T>
__global__ __launch_bounds__(256) void neptune_reduce_final(const T* __restrict__ partials, int n, T* __restrict__ out) {
  __shared__ T lds[4];
  T acc = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += partials[i];
  const T r = block_sum(acc, lds);
  if (threadIdx.x == 0) *out = r;
}

}  // namespace neptune_hip

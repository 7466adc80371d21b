// apply_common.hpp -- shared device/host helpers for the NeptuneIR apply kernels (gfx950).
//
// Vocabulary (follows the reference's dialect, include/Dialect/NeptuneIR/*.td):
//   temp / field : dense row-major buffer of shape ub-lb for a logical box [lb,ub)
//   apply        : out<p> = body(p, access(k,off) = in_k<p+off>) for p in apply.bounds,
//                  out[q] = in_0[q] elsewhere (copy-through)
//   access       : read of input k at a compile-time offset from the current point
//   body         : the scalar region of the apply, evaluated in textual op order,
//                  strict IEEE (build with -ffp-contract=off)
//
// A *body functor* is what the lowering emits for one apply region (and what
// builtin_bodies.hpp hand-writes for the committed fixtures):
//
//   struct Body {
//     template <class A> __device__ T operator()(const A& a) const {
//       T c = a.template get<0, 0, 0, 0>();      // access %in0[0,0,0]
//       T w = a.template get<0, 0, 0, -1>();     // access %in0[0,0,-1]
//       long i = a.template idx<0>();            // region index argument #0
//       ...
//     }
//   };
//
// and a *footprint* describes, at compile time, which offsets the body touches so the
// march kernel knows what to keep in registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

namespace neptune_hip {

constexpr int kWave = 64;   // gfx950 wavefront
constexpr int kMaxRank = 3;
constexpr int kMaxInputs = 4;
constexpr int kNumXcd = 8;  // MI355X: 8 XCDs, workgroups dealt round-robin over them

// ---- compile-time loop: indices stay constants, arrays stay in VGPRs -----------------
template <class F, int... Is>
__host__ __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__host__ __device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// ---- footprint ------------------------------------------------------------------------
// HALO_INPUT : index of the (lowest) input read at non-zero offsets, or -1 if every access is
//              at offset 0 (a pointwise apply); HALO_MASK has one bit per such input.
// R0,R1,R2   : max |offset| of the halo inputs along the march (I), row (J) and contiguous (K)
//              axes *after* the rank mapping below.
// BOX        : true if some access has more than one non-zero offset component (27-point);
//              false for star stencils (5/7-point), which need no corner data.
template <int HALO_INPUT_, int R0_, int R1_, int R2_, bool BOX_, bool MARCH_OK_ = true,
          unsigned HALO_MASK_ = (HALO_INPUT_ >= 0 ? (1u << (HALO_INPUT_ < 0 ? 0 : HALO_INPUT_)) : 0u)>
struct Footprint {
  static constexpr int HALO_INPUT = HALO_INPUT_;  // lowest halo input (or -1); kept for readability
  // bit k set: input k is read at non-zero offsets and gets a register ring of its own.  Several
  // halo inputs (e.g. the h and q fields of a shallow-water residual) share the radii below.
  static constexpr unsigned HALO_MASK = HALO_MASK_;
  static constexpr int R0 = R0_, R1 = R1_, R2 = R2_;
  static constexpr bool BOX = BOX_;
  static constexpr bool MARCH_OK = MARCH_OK_;
};
constexpr int popcount_u(unsigned m) { int c = 0; for (; m; m &= m - 1) ++c; return c; }
// index of input `in` among the set bits of `mask` (its ring slot)
constexpr int halo_slot(unsigned mask, int in) { return popcount_u(mask & ((1u << in) - 1u)); }
// input index of ring slot `h`
constexpr int halo_input_of(unsigned mask, int h) {
  for (int k = 0; k < 32; ++k)
    if (mask & (1u << k)) { if (h == 0) return k; --h; }
  return -1;
}

// Value of a cell outside apply.bounds: input 0 at the same physical index (copy-through,
// DataflowLowering.cpp:283-287), unless the body functor supplies `T outside(T through) const` --
// the fused explicit time step does, because outside its rhs operator's bounds the rhs IS the
// copy-through of the state and the axpy still applies there.
template <class Body, class T, class = void>
struct OutsideOf {
  static __device__ __forceinline__ T apply(const Body&, T through) { return through; }
};
template <class Body, class T>
struct OutsideOf<Body, T, std::void_t<decltype(std::declval<const Body&>().outside(std::declval<T>()))>> {
  static __device__ __forceinline__ T apply(const Body& b, T through) { return b.outside(through); }
};

// Rank mapping onto the kernel's (I, J, K) axes.  K is always the contiguous last dim, I the
// slowest.  rank 3: (d0,d1,d2) -> (I,J,K); rank 2: (d0,d1) -> (I,K), J has extent 1;
// rank 1: (d0) -> (K), I and J have extent 1.
template <int RANK> struct AxisMap;
template <> struct AxisMap<3> { static constexpr int I = 0, J = 1, K = 2; };
template <> struct AxisMap<2> { static constexpr int I = 0, J = -1, K = 1; };
template <> struct AxisMap<1> { static constexpr int I = -1, J = -1, K = 0; };

template <int RANK, int AXIS_DIM, int... O>
struct PickOffset {
  static constexpr int value = [] {
    constexpr int offs[sizeof...(O) ? sizeof...(O) : 1] = {O...};
    return (AXIS_DIM >= 0 && AXIS_DIM < (int)sizeof...(O)) ? offs[AXIS_DIM < 0 ? 0 : AXIS_DIM] : 0;
  }();
};

// ---- deterministic test/bench field: exact on host and device ------------------------
// splitmix64 finaliser -> 24 (f32) or 53 (f64) mantissa bits -> [-1, 1)
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ double hash_f64(int64_t index, uint64_t seed) {
  uint64_t h = mix64((uint64_t)index ^ mix64(seed));
  // 52 random bits -> [0,2) - 1 ; every step exact in binary64
  return (double)(h >> 12) * (1.0 / 2251799813685248.0) - 1.0;  // 2^-51
}
__host__ __device__ __forceinline__ float hash_f32(int64_t index, uint64_t seed) {
  uint64_t h = mix64((uint64_t)index ^ mix64(seed));
  return (float)(h >> 41) * (1.0f / 4194304.0f) - 1.0f;  // 23 bits * 2^-22
}

// ---- grids beyond 2^32 work-items --------------------------------------------------------------------
// HIP caps a grid dimension at 2^32 work-items (blocks x lanes), which a one-workgroup-per-256-cells launch
// over a 2048^3 field exceeds.  Such launches fold their workgroup count into (x, y).
constexpr uint32_t kMaxGridX = 1u << 22;  // x 256 lanes = 2^30 work-items along x
inline dim3 grid_for_blocks(int64_t blocks) {
  if (blocks <= (int64_t)kMaxGridX) return dim3((uint32_t)blocks);
  return dim3(kMaxGridX, (uint32_t)((blocks + kMaxGridX - 1) / kMaxGridX));
}
__device__ __forceinline__ int64_t linear_block() { return (int64_t)blockIdx.y * gridDim.x + blockIdx.x; }

// ---- XCD-aware workgroup id ------------------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  Remap so that
// each XCD owns one contiguous run of virtual ids: neighbouring tiles (which share halo
// rows) then hit the same L2.  Bijective for any grid size; speed only, never correctness.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nb) {
  const uint32_t q = nb / kNumXcd, r = nb % kNumXcd;
  const uint32_t xcd = b % kNumXcd, pos = b / kNumXcd;
  return xcd < r ? xcd * (q + 1) + pos : r * (q + 1) + (xcd - r) * q + pos;
}

}  // namespace neptune_hip

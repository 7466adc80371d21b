// body_ops.hpp -- device helpers for arith/math ops that have no single C++ operator.
// Semantics follow the MLIR arith dialect the reference's bodies are written in.
#pragma once
#include <hip/hip_runtime.h>

namespace neptune_hip {
namespace ops {

// arith.maximumf / minimumf: NaN if either operand is NaN; -0.0 < +0.0
template <class T>
__host__ __device__ __forceinline__ T maximumf(T a, T b) {
  if (a != a) return a;
  if (b != b) return b;
  if (a == b) return __builtin_signbit(a) ? b : a;
  return a > b ? a : b;
}
template <class T>
__host__ __device__ __forceinline__ T minimumf(T a, T b) {
  if (a != a) return a;
  if (b != b) return b;
  if (a == b) return __builtin_signbit(a) ? a : b;
  return a < b ? a : b;
}
// arith.maxnumf / minnumf: the non-NaN operand if exactly one is NaN
template <class T>
__host__ __device__ __forceinline__ T maxnumf(T a, T b) {
  if (a != a) return b;
  if (b != b) return a;
  return a > b ? a : b;
}
template <class T>
__host__ __device__ __forceinline__ T minnumf(T a, T b) {
  if (a != a) return b;
  if (b != b) return a;
  return a < b ? a : b;
}
__host__ __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
__host__ __device__ __forceinline__ float sqrt(float x) { return ::sqrtf(x); }
__host__ __device__ __forceinline__ double absf(double x) { return ::fabs(x); }
__host__ __device__ __forceinline__ float absf(float x) { return ::fabsf(x); }

// math.floor / ceil / copysign: exact
__host__ __device__ __forceinline__ double floor(double x) { return ::floor(x); }
__host__ __device__ __forceinline__ float floor(float x) { return ::floorf(x); }
__host__ __device__ __forceinline__ double ceil(double x) { return ::ceil(x); }
__host__ __device__ __forceinline__ float ceil(float x) { return ::ceilf(x); }
__host__ __device__ __forceinline__ double copysign(double a, double b) { return ::copysign(a, b); }
__host__ __device__ __forceinline__ float copysign(float a, float b) { return ::copysignf(a, b); }

// Elementary functions (math.exp / log / sin / cos / tanh / powf).  Unlike everything above these are NOT exactly
// specified: the reference lowers them to libm calls (math-to-llvm -> intrinsics -> libm on the CPU), this backend
// to the device math library; both are accurate to about an ulp, not correctly rounded, so results agree to a few
// ulp instead of bit for bit.  The lowering report flags bodies that use them ("exact": false).
#define NEPTUNE_ELEMENTARY(name, f64fn, f32fn)                                         \
  __host__ __device__ __forceinline__ double name(double x) { return f64fn(x); }       \
  __host__ __device__ __forceinline__ float name(float x) { return f32fn(x); }
NEPTUNE_ELEMENTARY(exp, ::exp, ::expf)
NEPTUNE_ELEMENTARY(log, ::log, ::logf)
NEPTUNE_ELEMENTARY(sin, ::sin, ::sinf)
NEPTUNE_ELEMENTARY(cos, ::cos, ::cosf)
NEPTUNE_ELEMENTARY(tanh, ::tanh, ::tanhf)
#undef NEPTUNE_ELEMENTARY
__host__ __device__ __forceinline__ double powf(double a, double b) { return ::pow(a, b); }
__host__ __device__ __forceinline__ float powf(float a, float b) { return ::powf(a, b); }

// Explicit time_advance: out = s + dt * k with the two roundings the reference's lowering produces
// (arith.mulf then arith.addf, HighLevelConvertion.cpp:109-110).  Input 0 = state, input 1 = rhs(state).
template <class T, int RANK>
struct EulerAxpy {
  T dt;
  template <class A>
  __device__ __forceinline__ T operator()(const A& a) const {
    T s0, k0;
    if constexpr (RANK == 1) { s0 = a.template get<0, 0>(); k0 = a.template get<1, 0>(); }
    else if constexpr (RANK == 2) { s0 = a.template get<0, 0, 0>(); k0 = a.template get<1, 0, 0>(); }
    else { s0 = a.template get<0, 0, 0, 0>(); k0 = a.template get<1, 0, 0, 0>(); }
    const T dt_k = dt * k0;
    return s0 + dt_k;
  }
};

// The same step in ONE pass over the state when the rhs opdef is a single apply of the state:
// inside that apply's bounds k = Rhs(a); outside them the rhs result is its copy-through of the
// state, so k = s (see OutsideOf in apply_common.hpp).  Same two roundings as EulerAxpy, hence the
// same bits as the two-kernel form -- it only saves the round trip of k through HBM
// (2 field passes instead of 5).
template <class Rhs, class T, int RANK>
struct EulerFused {
  T dt;
  template <class A>
  __device__ __forceinline__ T operator()(const A& a) const {
    const T k0 = Rhs{}(a);
    T s0;
    if constexpr (RANK == 1) s0 = a.template get<0, 0>();
    else if constexpr (RANK == 2) s0 = a.template get<0, 0, 0>();
    else s0 = a.template get<0, 0, 0, 0>();
    const T dt_k = dt * k0;
    return s0 + dt_k;
  }
  __device__ __forceinline__ T outside(T s0) const {
    const T dt_k = dt * s0;
    return s0 + dt_k;
  }
};

}  // namespace ops
}  // namespace neptune_hip

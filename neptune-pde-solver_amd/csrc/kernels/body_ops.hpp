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

}  // namespace ops
}  // namespace neptune_hip

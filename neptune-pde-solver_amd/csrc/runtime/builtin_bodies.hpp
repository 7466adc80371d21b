// builtin_bodies.hpp -- body functors of the committed fixtures, written exactly as the
// lowering emits them (one statement per IR op, textual order, no reassociation).  They let
// the runtime library, the bench and the parity tests exercise the kernels without running
// the lowering + hipcc first.  tests/test_lowering.py checks that the emitter's output for each
// fixture is statement-for-statement this code.
#pragma once
#include "../kernels/apply_common.hpp"

namespace neptune_hip {
namespace builtin {

// tests/mlir_tests/conversion_tests/apply-2d-5pt.mlir  (@lap2d)
struct Lap2D5 {
  using T = double;
  static constexpr int RANK = 2, NIN = 1;
  using FP = Footprint</*halo input*/ 0, /*R0*/ 1, /*R1*/ 0, /*R2*/ 1, /*box*/ false>;
  static constexpr int32_t radius[kMaxInputs][kMaxRank] = {{1, 1, 0}, {-1, -1, -1}, {-1, -1, -1}, {-1, -1, -1}};
  template <class A>
  __device__ __forceinline__ double operator()(const A& a) const {
    const double c = a.template get<0, 0, 0>();
    const double n = a.template get<0, -1, 0>();
    const double s = a.template get<0, 1, 0>();
    const double w = a.template get<0, 0, -1>();
    const double e = a.template get<0, 0, 1>();
    const double four = 4.0;
    const double dxinv2 = 0.125;
    const double t0 = n + s;
    const double t1 = t0 + w;
    const double t2 = t1 + e;
    const double t3 = four * c;
    const double t4 = t2 - t3;
    const double lap = dxinv2 * t4;
    return lap;
  }
};

// tests/mlir_tests/conversion_tests/apply-3d-7pt.mlir  (@lap3d)
struct Lap3D7 {
  using T = double;
  static constexpr int RANK = 3, NIN = 1;
  using FP = Footprint<0, 1, 1, 1, false>;
  static constexpr int32_t radius[kMaxInputs][kMaxRank] = {{1, 1, 1}, {-1, -1, -1}, {-1, -1, -1}, {-1, -1, -1}};
  template <class A>
  __device__ __forceinline__ double operator()(const A& a) const {
    const double c = a.template get<0, 0, 0, 0>();
    const double xm = a.template get<0, -1, 0, 0>();
    const double xp = a.template get<0, 1, 0, 0>();
    const double ym = a.template get<0, 0, -1, 0>();
    const double yp = a.template get<0, 0, 1, 0>();
    const double zm = a.template get<0, 0, 0, -1>();
    const double zp = a.template get<0, 0, 0, 1>();
    const double six = 6.0;
    const double dxinv2 = 0.0625;
    const double t0 = xm + xp;
    const double t1 = t0 + ym;
    const double t2 = t1 + yp;
    const double t3 = t2 + zm;
    const double t4 = t3 + zp;
    const double t5 = six * c;
    const double t6 = t4 - t5;
    const double lap = dxinv2 * t6;
    return lap;
  }
};

// tests/mlir_tests/conversion_tests/apply-3d-27pt.mlir  (@lap27)
struct Lap3D27 {
  using T = float;
  static constexpr int RANK = 3, NIN = 1;
  using FP = Footprint<0, 1, 1, 1, true>;
  static constexpr int32_t radius[kMaxInputs][kMaxRank] = {{1, 1, 1}, {-1, -1, -1}, {-1, -1, -1}, {-1, -1, -1}};
  template <class A>
  __device__ __forceinline__ float operator()(const A& a) const {
    const float c = a.template get<0, 0, 0, 0>();
    const float ammm = a.template get<0, -1, -1, -1>();
    const float ammz = a.template get<0, -1, -1, 0>();
    const float ammp = a.template get<0, -1, -1, 1>();
    const float amzm = a.template get<0, -1, 0, -1>();
    const float amzz = a.template get<0, -1, 0, 0>();
    const float amzp = a.template get<0, -1, 0, 1>();
    const float ampm = a.template get<0, -1, 1, -1>();
    const float ampz = a.template get<0, -1, 1, 0>();
    const float ampp = a.template get<0, -1, 1, 1>();
    const float azmm = a.template get<0, 0, -1, -1>();
    const float azmz = a.template get<0, 0, -1, 0>();
    const float azmp = a.template get<0, 0, -1, 1>();
    const float azzm = a.template get<0, 0, 0, -1>();
    const float azzp = a.template get<0, 0, 0, 1>();
    const float azpm = a.template get<0, 0, 1, -1>();
    const float azpz = a.template get<0, 0, 1, 0>();
    const float azpp = a.template get<0, 0, 1, 1>();
    const float apmm = a.template get<0, 1, -1, -1>();
    const float apmz = a.template get<0, 1, -1, 0>();
    const float apmp = a.template get<0, 1, -1, 1>();
    const float apzm = a.template get<0, 1, 0, -1>();
    const float apzz = a.template get<0, 1, 0, 0>();
    const float apzp = a.template get<0, 1, 0, 1>();
    const float appm = a.template get<0, 1, 1, -1>();
    const float appz = a.template get<0, 1, 1, 0>();
    const float appp = a.template get<0, 1, 1, 1>();
    const float c26 = 26.0f;
    const float dxinv2 = 0.015625f;
    const float s0 = ammm + ammz;
    const float s1 = s0 + ammp;
    const float s2 = s1 + amzm;
    const float s3 = s2 + amzz;
    const float s4 = s3 + amzp;
    const float s5 = s4 + ampm;
    const float s6 = s5 + ampz;
    const float s7 = s6 + ampp;
    const float s8 = s7 + azmm;
    const float s9 = s8 + azmz;
    const float s10 = s9 + azmp;
    const float s11 = s10 + azzm;
    const float s12 = s11 + azzp;
    const float s13 = s12 + azpm;
    const float s14 = s13 + azpz;
    const float s15 = s14 + azpp;
    const float s16 = s15 + apmm;
    const float s17 = s16 + apmz;
    const float s18 = s17 + apmp;
    const float s19 = s18 + apzm;
    const float s20 = s19 + apzz;
    const float s21 = s20 + apzp;
    const float s22 = s21 + appm;
    const float s23 = s22 + appz;
    const float s24 = s23 + appp;
    const float t0 = c26 * c;
    const float t1 = s24 - t0;
    const float lap = dxinv2 * t1;
    return lap;
  }
};

// @ac_lap of the reference's test/smoke_tests/smoke_time_advance.mlir:13-29
// (the 1-D input every reference smoke test is built around)
struct Lap1D3 {
  using T = double;
  static constexpr int RANK = 1, NIN = 1;
  using FP = Footprint<0, 0, 0, 1, false>;
  static constexpr int32_t radius[kMaxInputs][kMaxRank] = {{1, 0, 0}, {-1, -1, -1}, {-1, -1, -1}, {-1, -1, -1}};
  template <class A>
  __device__ __forceinline__ double operator()(const A& a) const {
    const double um1 = a.template get<0, -1>();
    const double u0 = a.template get<0, 0>();
    const double up1 = a.template get<0, 1>();
    const double two = 2.0;
    const double dxinv2 = 100.0;
    const double t0 = two * u0;
    const double t1 = um1 - t0;
    const double t2 = t1 + up1;
    const double lap_i = dxinv2 * t2;
    return lap_i;
  }
};

}  // namespace builtin
}  // namespace neptune_hip

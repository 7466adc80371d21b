// apply_launch.hpp -- host side of one `neptune_ir.apply`: validate the geometry, choose a
// kernel, fill its parameters, hipLaunchKernelGGL.  Shared by the runtime library (built-in
// bodies) and by every module the lowering emits (generated bodies).
//
// Plan rules (all decided on the host, before anything runs on the device):
//   * geometry must be well formed and every access of every in-bounds point must stay inside
//     its input's box -- the reference performs no check there and reads out of bounds
//     (lib/Passes/DataflowLowering.cpp:380-410; test/smoke_tests/smoke_apply.mlir:4-9 does);
//     this backend refuses such a plan instead of emulating undefined behaviour.
//   * march kernel when: footprint allows it, input 0 shares the result's box and every other input's box CONTAINS it
//     (staggered grids, fields that carry their ghost layers: MarchParams::view), the contiguous extent is at least one
//     wave wide, the result and the inputs in the result's box are 16-byte aligned, and the region only restricts dim 0.
//     A march tile index may stand for the plane-in-LDS kernels (apply_plane.hpp): tile 7 for every footprint they can run,
//     every index for footprints nothing else holds (3-D stars beyond radius 4, several wide halo inputs, radius-2 boxes).
//   * otherwise the direct kernel.
#pragma once
#include <array>
#include <map>
#include <math.h>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <typeinfo>

#include "../../../include/neptune_hip.h"
#include "apply_common.hpp"
#include "apply_direct.hpp"
#include "apply_march.hpp"
#include "apply_plane.hpp"

#ifndef NEPTUNE_HIP_FULL_VARIANTS
#define NEPTUNE_HIP_FULL_VARIANTS 0  // generated modules compile the default tile only
#endif

namespace neptune_hip {

#define NEPTUNE_HIP_CHECK(expr)                                                              \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      fprintf(stderr, "[NeptuneRT][HIP] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), \
              __FILE__, __LINE__);                                                           \
      abort();                                                                               \
    }                                                                                        \
  } while (0)

// ---- geometry checks ------------------------------------------------------------------
inline int geom_validate(const neptune_hip_apply_geom_t* g) {
  if (!g) return NEPTUNE_HIP_EINVAL;
  if (g->rank < 1 || g->rank > kMaxRank) return NEPTUNE_HIP_EINVAL;
  if (g->num_inputs < 1 || g->num_inputs > kMaxInputs) return NEPTUNE_HIP_EINVAL;
  // bounds that are empty along ANY dimension make the reference's scf.for nest run zero trips
  // (DataflowLowering.cpp:289-310): nothing is stored, so where the box lies along the other dimensions is irrelevant
  bool empty = false;
  for (int d = 0; d < g->rank; ++d) empty = empty || g->lb[d] >= g->ub[d];
  for (int d = 0; d < g->rank; ++d) {
    const int64_t n = g->out_ub[d] - g->out_lb[d];
    if (n <= 0) return NEPTUNE_HIP_EINVAL;
    // result shape must equal input 0's shape (cast at DataflowLowering.cpp:285-286)
    if (g->in_ub[0][d] - g->in_lb[0][d] != n) return NEPTUNE_HIP_EINVAL;
    for (int k = 0; k < g->num_inputs; ++k)
      if (g->in_ub[k][d] - g->in_lb[k][d] <= 0) return NEPTUNE_HIP_EINVAL;
    if (g->lb[d] > g->ub[d]) return NEPTUNE_HIP_EINVAL;
    // the yielded scalar is stored at p - out_lb (:427-444): bounds must lie in the result box
    if (!empty && (g->lb[d] < g->out_lb[d] || g->ub[d] > g->out_ub[d])) return NEPTUNE_HIP_EOOB;
    if (g->region_lb[d] < 0 || g->region_ub[d] > n || g->region_lb[d] > g->region_ub[d])
      return NEPTUNE_HIP_EINVAL;
  }
  return NEPTUNE_HIP_OK;
}

inline bool geom_bounds_empty(const neptune_hip_apply_geom_t* g) {
  for (int d = 0; d < g->rank; ++d)
    if (g->lb[d] >= g->ub[d]) return true;
  return false;
}

// What an apply's UNCONDITIONAL accesses reach per input and dimension, as the offsets themselves: lo = the most negative
// offset (<= 0), hi = the most positive (>= 0); hi < lo: the input is not accessed unconditionally.  A staggered-grid body
// reads a face field at [0] and [+1] only -- its reach is (0, +1), and bounds that start at the face field's lower bound
// are legal (the symmetric radius form below would refuse them).  The lowering emits one per apply.
struct Reach {
  int32_t lo[NEPTUNE_HIP_MAX_INPUTS][NEPTUNE_HIP_MAX_RANK];
  int32_t hi[NEPTUNE_HIP_MAX_INPUTS][NEPTUNE_HIP_MAX_RANK];
};
inline int geom_check_radius(const neptune_hip_apply_geom_t* g, const Reach& reach) {
  int rc = geom_validate(g);
  if (rc != NEPTUNE_HIP_OK) return rc;
  if (geom_bounds_empty(g)) return NEPTUNE_HIP_OK;
  for (int k = 0; k < g->num_inputs; ++k)
    for (int d = 0; d < g->rank; ++d) {
      const int64_t lo = reach.lo[k][d], hi = reach.hi[k][d];
      if (hi < lo) continue;  // input not accessed at all
      if (g->lb[d] + lo < g->in_lb[k][d] || g->ub[d] + hi > g->in_ub[k][d]) return NEPTUNE_HIP_EOOB;
    }
  return NEPTUNE_HIP_OK;
}

inline int geom_check_radius(const neptune_hip_apply_geom_t* g,
                             const int32_t radius[NEPTUNE_HIP_MAX_INPUTS][NEPTUNE_HIP_MAX_RANK]) {
  int rc = geom_validate(g);
  if (rc != NEPTUNE_HIP_OK) return rc;
  if (geom_bounds_empty(g)) return NEPTUNE_HIP_OK;
  for (int k = 0; k < g->num_inputs; ++k)
    for (int d = 0; d < g->rank; ++d) {
      const int64_t r = radius[k][d];
      if (r < 0) continue;  // input not accessed at all
      if (g->lb[d] - r < g->in_lb[k][d] || g->ub[d] + r > g->in_ub[k][d]) return NEPTUNE_HIP_EOOB;
    }
  return NEPTUNE_HIP_OK;
}

// logical-dim arrays -> kernel axis order (I,J,K); absent axes get `fill`
template <int RANK>
inline void to_axes(const int64_t* src, int64_t dst[3], int64_t fill) {
  dst[0] = dst[1] = dst[2] = fill;
  if constexpr (RANK == 3) { dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; }
  else if constexpr (RANK == 2) { dst[0] = src[0]; dst[2] = src[1]; }
  else { dst[2] = src[0]; }
}

// ---- march tile variants --------------------------------------------------------------
struct MarchVariant {
  int RJ, WJ, WK;
  bool dpp, nt;
  int PF;
  bool ntl, ldsj, jk, jhl;
  int KD;
  bool pln;  // the plane-in-LDS kernel (apply_plane.hpp) for footprints it can run; the same tile of the march kernel otherwise
  const char* name;
};
// rank 3: a workgroup is WJ x WK waves, each lane owns RJ rows x 16 B, PF planes in flight.
// X(index, RJ, WJ, WK, DPP, NT, PF, NTL, name)
// The first variants of each rank (NEPTUNE_MARCH*_DEFAULT) are the defaults and are compiled into every lowered
// module; the others stay in the runtime library for tools/sweep.py and for the parity tests,
// which run every one of them.  Measured on MI355X (profiles/r01_sweep_report.txt):
//   rank 3, star  : 0  rj4_wj4_wk2_pf2_lds   (1024^3 fp64 7-point: 6.1 TB/s)
//   rank 3, box   : 1  rj4_wj8_wk1_pf2_lds   (512^3 fp32 27-point: 5.6 TB/s)
//   rank 3, star of radius 2 : 2  rj2_wj8_wk1_pf2_lds_jhl  (5 live planes: fewer rows per lane,
//                      J halo rows only for the centre plane, so the tile fits the register file)
//   rank 3, star of radius 4 : 5  rj2_wj8_wk1_pf1_lds_jhl  (9 live planes: one plane in flight instead of two keeps
//                      the 25-point operator out of scratch: 512^3 fp64 2.7 TB/s against 2.1; at radius 2-3 the deeper
//                      prefetch of tile 2 wins, 3.6 against 2.8 for the 19-point operator)
//   rank 3, large star fields: 4  rj4_wj16_wk1_pf1_lds (64 rows x one wave span, 16 waves: within 1 % of tile 0 on
//                      the fastest boxes of the pool and 4-10 % ahead on the others, 512^3-1024^3)
//   rank 3, largest star fields (two rounds of workgroups and more): 6  rj8_wj4_wk2_pf1_lds_jhl_kd2 (32 rows x two wave
//                      spans, EIGHT rows per lane, 8 waves = one workgroup per CU at ~200 VGPRs, K halos requested with
//                      the rows of the same plane: 1024^3 fp64 7-point +5 % over tile 4 on the same box, HBM-side traffic
//                      1.02x algorithmic against 1.08x -- profiles/r02_headline_search.txt)
//   rank 3, small fields     : 3  rj2_wj4_wk1_pf2  (8 rows x one wave span, 4 waves, no LDS: the
//                      problem is cut into 4-8x more workgroups; 128^3: 10 us instead of 20,
//                      27-point 256^3: 28 us instead of 46, profiles/r01_size_sweep.txt)
//   rank 2        : 0  tile_rj4_wj8_wk1      (8192^2 fp64 5-point: 6.15 TB/s),
//                   2  tile_rj4_wj4_wk1      (two halo inputs: 5.6 TB/s against 4.9 on the march form), and
//                   1  wk4_pf4, the march form (fields of 2 GiB and more, several halo inputs)
// X(index, RJ, WJ, WK, DPP, NT, PF, NTL, LDSJ, JK, JHL, KD, PLN, name)
#define NEPTUNE_MARCH3_DEFAULT(X)                                                   \
  X(0, 4, 4, 2, true, true, 2, false, true, false, false, 1, false, "rj4_wj4_wk2_pf2_lds")    \
  X(1, 4, 8, 1, true, true, 2, false, true, false, false, 1, false, "rj4_wj8_wk1_pf2_lds")    \
  X(2, 2, 8, 1, true, true, 2, false, true, false, true, 1, false, "rj2_wj8_wk1_pf2_lds_jhl") \
  X(3, 2, 4, 1, true, true, 2, false, false, false, false, 1, false, "rj2_wj4_wk1_pf2")      \
  X(4, 4, 16, 1, true, true, 1, false, true, false, false, 1, false, "rj4_wj16_wk1_pf1_lds")  \
  X(5, 2, 8, 1, true, true, 1, false, true, false, true, 1, false, "rj2_wj8_wk1_pf1_lds_jhl") \
  X(6, 8, 4, 2, true, true, 1, false, true, false, true, 2, false, "rj8_wj4_wk2_pf1_lds_jhl_kd2") \
  X(7, 4, 8, 1, true, true, 1, false, true, false, true, 1, true, "pln_rj4_wj8_wk1_pf1")
#define NEPTUNE_MARCH2_DEFAULT(X)                                            \
  X(0, 4, 8, 1, true, true, 1, false, true, true, false, 1, false, "tile_rj4_wj8_wk1")        \
  X(1, 1, 1, 4, true, true, 4, false, false, false, false, 1, false, "wk4_pf4")             \
  X(2, 4, 4, 1, true, true, 1, false, true, true, false, 1, false, "tile_rj4_wj4_wk1")
#if NEPTUNE_HIP_FULL_VARIANTS
#define NEPTUNE_MARCH3_VARIANTS(X)                \
  NEPTUNE_MARCH3_DEFAULT(X)                                                   \
  X(8, 4, 4, 1, true, true, 2, false, false, false, false, 1, false, "rj4_wj4_wk1_pf2") \
  X(9, 4, 4, 1, false, false, 1, false, false, false, false, 1, false, "rj4_wj4_wk1_pf1_shfl_plainst") \
  X(10, 8, 2, 1, true, true, 1, false, false, false, false, 1, false, "rj8_wj2_wk1_pf1") \
  X(11, 4, 4, 1, true, true, 1, false, false, false, false, 1, false, "rj4_wj4_wk1_pf1") \
  X(12, 4, 4, 1, true, true, 2, false, true, false, false, 1, false, "rj4_wj4_wk1_pf2_lds") \
  X(13, 2, 8, 1, true, true, 4, false, true, false, false, 1, false, "rj2_wj8_wk1_pf4_lds") \
  X(14, 8, 4, 1, true, true, 1, false, true, false, false, 1, false, "rj8_wj4_wk1_pf1_lds") \
  X(15, 8, 4, 2, true, true, 1, false, true, false, false, 1, false, "rj8_wj4_wk2_pf1_lds") \
  X(16, 4, 4, 2, true, true, 3, false, true, false, false, 1, false, "rj4_wj4_wk2_pf3_lds") \
  X(17, 4, 2, 4, true, true, 2, false, true, false, false, 1, false, "rj4_wj2_wk4_pf2_lds") \
  X(18, 2, 16, 1, true, true, 3, false, true, false, false, 1, false, "rj2_wj16_wk1_pf3_lds") \
  X(19, 4, 4, 2, true, true, 2, false, true, false, true, 1, false, "rj4_wj4_wk2_pf2_lds_jhl") \
  X(20, 4, 8, 1, true, true, 2, false, true, false, true, 1, false, "rj4_wj8_wk1_pf2_lds_jhl") \
  X(21, 8, 4, 2, true, true, 1, false, true, false, true, 1, false, "rj8_wj4_wk2_pf1_lds_jhl") \
  X(22, 4, 4, 2, true, true, 3, false, true, false, true, 1, false, "rj4_wj4_wk2_pf3_lds_jhl") \
  X(23, 8, 4, 1, true, true, 1, false, true, false, true, 1, false, "rj8_wj4_wk1_pf1_lds_jhl") \
  X(24, 8, 4, 2, true, true, 2, false, true, false, true, 1, false, "rj8_wj4_wk2_pf2_lds_jhl") \
  X(25, 4, 4, 2, true, true, 4, false, true, false, true, 1, false, "rj4_wj4_wk2_pf4_lds_jhl") \
  X(26, 4, 16, 1, true, true, 2, false, true, false, true, 1, false, "rj4_wj16_wk1_pf2_lds_jhl") \
  X(27, 2, 16, 1, true, true, 3, false, true, false, true, 1, false, "rj2_wj16_wk1_pf3_lds_jhl") \
  X(28, 4, 8, 2, true, true, 2, false, true, false, true, 1, false, "rj4_wj8_wk2_pf2_lds_jhl") \
  X(29, 4, 8, 2, true, true, 1, false, true, false, false, 1, false, "rj4_wj8_wk2_pf1_lds") \
  X(30, 4, 4, 2, true, true, 2, false, true, false, false, 2, false, "rj4_wj4_wk2_pf2_lds_kd2") \
  X(31, 4, 4, 2, true, true, 2, false, true, false, false, 3, false, "rj4_wj4_wk2_pf2_lds_kd3") \
  X(32, 8, 4, 2, true, true, 1, false, true, false, false, 2, false, "rj8_wj4_wk2_pf1_lds_kd2") \
  X(33, 8, 8, 1, true, true, 1, false, true, false, false, 2, false, "rj8_wj8_wk1_pf1_lds_kd2") \
  X(34, 8, 2, 4, true, true, 1, false, true, false, false, 2, false, "rj8_wj2_wk4_pf1_lds_kd2") \
  X(35, 4, 16, 1, true, true, 1, false, true, false, false, 2, false, "rj4_wj16_wk1_pf1_lds_kd2") \
  X(36, 4, 16, 1, true, true, 1, true, true, false, false, 1, false, "rj4_wj16_wk1_pf1_lds_ntl") \
  X(37, 4, 8, 1, true, true, 2, false, true, false, true, 1, true, "pln_rj4_wj8_wk1_pf2") \
  X(38, 4, 4, 2, true, true, 1, false, true, false, true, 1, true, "pln_rj4_wj4_wk2_pf1") \
  X(39, 2, 8, 1, true, true, 2, false, true, false, true, 1, true, "pln_rj2_wj8_wk1_pf2") \
  X(40, 4, 4, 1, true, true, 1, false, true, false, true, 1, true, "pln_rj4_wj4_wk1_pf1") \
  X(41, 2, 12, 1, true, true, 1, false, true, false, true, 1, true, "pln_rj2_wj12_wk1_pf1") \
  X(42, 2, 16, 1, true, true, 1, false, true, false, true, 1, true, "pln_rj2_wj16_wk1_pf1") \
  X(43, 1, 16, 1, true, true, 2, false, true, false, true, 1, true, "pln_rj1_wj16_wk1_pf2") \
  X(44, 2, 16, 1, true, true, 2, false, true, false, true, 1, true, "pln_rj2_wj16_wk1_pf2") \
  X(45, 2, 16, 1, true, true, 3, false, true, false, true, 1, true, "pln_rj2_wj16_wk1_pf3") \
  X(46, 3, 8, 1, true, true, 2, false, true, false, true, 1, true, "pln_rj3_wj8_wk1_pf2")
#define NEPTUNE_MARCH2_VARIANTS(X)                \
  NEPTUNE_MARCH2_DEFAULT(X)                                      \
  X(3, 1, 1, 4, true, true, 2, false, false, false, false, 1, false, "wk4_pf2")  \
  X(4, 1, 1, 1, true, true, 4, false, false, false, false, 1, false, "wk1_pf4")  \
  X(5, 1, 1, 4, true, true, 8, false, false, false, false, 1, false, "wk4_pf8")  \
  X(6, 4, 16, 1, true, true, 1, false, true, true, false, 1, false, "tile_rj4_wj16_wk1")  \
  X(7, 8, 8, 1, true, true, 1, false, true, true, false, 1, false, "tile_rj8_wj8_wk1")    \
  X(8, 2, 16, 1, true, true, 1, false, true, true, false, 1, false, "tile_rj2_wj16_wk1")  \
  X(9, 4, 4, 2, true, true, 1, false, true, true, false, 1, false, "tile_rj4_wj4_wk2")    \
  X(10, 4, 8, 2, true, true, 1, false, true, true, false, 1, false, "tile_rj4_wj8_wk2")   \
  X(11, 8, 4, 1, true, true, 1, false, true, true, false, 1, false, "tile_rj8_wj4_wk1")   \
  X(12, 1, 1, 4, false, false, 1, false, false, false, false, 1, false, "wk4_pf1_shfl_plainst") \
  X(13, 4, 8, 1, false, false, 1, false, true, true, false, 1, false, "tile_rj4_wj8_wk1_shfl_plainst")
#else
#define NEPTUNE_MARCH3_VARIANTS(X) NEPTUNE_MARCH3_DEFAULT(X)
#define NEPTUNE_MARCH2_VARIANTS(X) NEPTUNE_MARCH2_DEFAULT(X)
#endif

#define NEPTUNE_MV_ROW(idx, RJ, WJ, WK, DPP, NT, PF, NTL, LDSJ, JK, JHL, KD, PLN, name) {RJ, WJ, WK, DPP, NT, PF, NTL, LDSJ, JK, JHL, KD, PLN, name},
constexpr MarchVariant kMarch3[] = {NEPTUNE_MARCH3_VARIANTS(NEPTUNE_MV_ROW)};
constexpr MarchVariant kMarch2[] = {NEPTUNE_MARCH2_VARIANTS(NEPTUNE_MV_ROW)};
#undef NEPTUNE_MV_ROW
// the tables are indexed by position, the launch switch by the X index: they must agree
#define NEPTUNE_MV_IDX(idx, RJ, WJ, WK, DPP, NT, PF, NTL, LDSJ, JK, JHL, KD, PLN, name) idx,
constexpr int kMarch3Idx[] = {NEPTUNE_MARCH3_VARIANTS(NEPTUNE_MV_IDX)};
constexpr int kMarch2Idx[] = {NEPTUNE_MARCH2_VARIANTS(NEPTUNE_MV_IDX)};
#undef NEPTUNE_MV_IDX
template <int N>
constexpr bool consecutive_from_zero(const int (&a)[N]) {
  for (int i = 0; i < N; ++i)
    if (a[i] != i) return false;
  return true;
}
static_assert(consecutive_from_zero(kMarch3Idx) && consecutive_from_zero(kMarch2Idx), "march variant lists must be listed in index order");
constexpr int kNumMarch3 = sizeof(kMarch3) / sizeof(kMarch3[0]);
constexpr int kNumMarch2 = sizeof(kMarch2) / sizeof(kMarch2[0]);

// rank 1: a single tile -- the field is one row, a workgroup is 4 waves side by side, every wave
// loads its 1 KiB, shifts, stores and retires (the access pattern of the fastest copy kernel)
constexpr MarchVariant kMarch1[] = {{1, 1, 4, true, true, 1, false, false, false, false, 1, false, "row_wk4"}};
inline int march_variant_count(int rank) { return rank == 3 ? kNumMarch3 : (rank == 2 ? kNumMarch2 : (rank == 1 ? 1 : 0)); }
inline const MarchVariant* march_variant(int rank, int v) {
  if (v < 0 || v >= march_variant_count(rank)) return nullptr;
  return rank == 3 ? &kMarch3[v] : (rank == 2 ? &kMarch2[v] : &kMarch1[v]);
}

// ragged rows: a stored cell's K neighbours must come from lanes that hold real cells, so a K radius of n lane vectors
// leaves n-1 more vectors per row to the tail launch
template <class FP, int VK>
constexpr int ragged_extra_vectors() { return FP::R2 > VK ? (FP::R2 + VK - 1) / VK - 1 : 0; }

// rows per lane a tile really gets for a footprint: rank-3 footprints with wide register state (boxes, radius 2 and
// more, several halo inputs) are capped at 4 -- on the 8-row tiles they would only spill
template <class FP, int RANK>
constexpr int march_rows(int rj) {
  return (RANK == 3 && rj > 4 && (FP::BOX || FP::R0 > 1 || FP::R1 > 1 || popcount_u(FP::HALO_MASK) > 1)) ? 4 : rj;
}

// the tile a table row becomes for a footprint: PLN rows run the plane kernel where it can (with the rows per lane its ring
// leaves room for) and fall back to the same shape of the march kernel elsewhere; footprints only the plane kernel can
// hold (radius beyond 4) run it on EVERY row
template <class T, class FP, int RANK>
constexpr bool plane_only() {
  constexpr int NH = popcount_u(FP::HALO_MASK);
  constexpr bool wide = FP::R0 > 1 || FP::R1 > 1 || FP::R2 > 1;
  // what the march kernel's registers cannot hold: stars beyond radius 4, two inputs read at offsets beyond radius 1, three or more
  return (plane_capable<FP, RANK>() && (FP::R0 > 4 || FP::R1 > 4 || FP::R2 > 4 || (NH > 1 && wide) || NH > 2)) ||
         (planes_capable<T, FP, RANK>() && (FP::R0 > 1 || FP::R1 > 1 || FP::R2 > 1));
}
// rank 2: footprints beyond the march kernel's registers (stars beyond radius 4 or a K radius beyond two lane vectors, boxes beyond
// radius 2, several inputs read at wide offsets) run the LDS tile kernel (neptune_apply_tile2) on every row
template <class T, class FP, int RANK>
constexpr bool tile2_only() {
  constexpr int VK = 16 / (int)sizeof(T), NH = popcount_u(FP::HALO_MASK), rbig = FP::R0 > FP::R2 ? FP::R0 : FP::R2;
  // ... and what they hold worse than the LDS tile does (8192^2, profiles/r02_plane.txt section 7): stars from radius 3 on (radius 3
  // ragged 4.76 TB/s against 4.07 on the march form, radius 4 level), 5x5 boxes (4.04 against 3.37)
  return tile2_capable<T, FP, RANK>() &&
         ((!FP::BOX && rbig > 2) || FP::R2 > 2 * VK || (FP::BOX && rbig > 1) || (NH > 1 && (rbig > 2 || (FP::BOX && rbig > 1))));
}
template <class T, class FP, int RANK, int RJ, int WJ, int WK, bool DPP, bool NT, int PF, bool NTL, bool LDSJ, bool JK, bool JHL, int KD, bool PLN>
struct TileFor {
  static constexpr bool star = plane_capable<FP, RANK>(), box = planes_capable<T, FP, RANK>(), flat = tile2_only<T, FP, RANK>();
  static constexpr bool pln = flat || ((star || box) && (PLN || plane_only<T, FP, RANK>()));
  // a march row that only lands here because nothing else can hold the footprint becomes one of TWO plane tiles (the 8-wave
  // default, or the 4-wave one for the small-field rows): eight different plane kernels per apply would only cost compile time
  static constexpr bool canon = pln && !PLN;
  static constexpr int wj = flat ? 8 : canon ? (WJ * WK <= 4 ? 4 : 8) : WJ, wk = (canon || flat) ? 1 : WK, rj = (canon || flat) ? 4 : RJ;
  static constexpr int rows = !pln ? march_rows<FP, RANK>(RJ) : flat ? tile2_rows<T, FP>(rj, wj, wk) : star ? plane_rows<T, FP>(rj, wj, wk) : planes_rows<T, FP>(rj, wj, wk);
  // several inputs read at offsets on the plane-in-LDS kernel: two planes in flight per input (radius-4 pair 512^3: 2.33 TB/s
  // against 1.93 with one; profiles/r03_highorder_tiles.txt)
  static constexpr int pf_star = (pln && !flat && star && popcount_u(FP::HALO_MASK) >= 2) ? 2 : 1;
  using type = std::conditional_t<flat, Tile<rows, 8, 1, true, true, 1, false, true, true, true, 1, true>,
               std::conditional_t<canon, Tile<rows, wj, 1, true, true, pf_star, false, true, false, true, 1, true>,
                                  Tile<rows, WJ, WK, DPP, NT, (pln && PF < pf_star) ? pf_star : PF, NTL, LDSJ, (JK) && RANK == 2, JHL, KD, pln>>>;
};
// star footprints: the centre plane in LDS, the ring of own cells in registers; box footprints: every live plane in LDS
template <class Body, class T, int RANK, int NIN, class FP, class TL>
constexpr auto march_kernel_fn() {
  if constexpr (TL::PLN && RANK == 2) return &neptune_apply_tile2<Body, T, NIN, FP, TL>;
  else if constexpr (TL::PLN && FP::BOX) return &neptune_apply_planes<Body, T, NIN, FP, TL>;
  else if constexpr (TL::PLN) return &neptune_apply_plane<Body, T, NIN, FP, TL>;
  else return &neptune_apply_march<Body, T, RANK, NIN, FP, TL>;
}

template <class Body, class T, int RANK, int NIN, class FP, class TL>
inline void launch_march_variant(MarchParams<T, NIN>& P, const Body& body, int64_t planes, int chunk_req,
                                 hipStream_t stream) {
  constexpr int VK = 16 / sizeof(T);
  constexpr int WJ = TL::WJ, WK = TL::WK, RJ = TL::RJ;
  const int64_t tileK = (int64_t)WK * kWave * VK, tileJ = (int64_t)WJ * RJ;
  P.Kl = P.N2 / VK * VK - VK;
  // a K radius beyond one vector reads two lanes to the right: leave one more vector of ragged rows to the tail
  P.Ks = (P.N2 % VK == 0) ? P.N2 : P.Kl - ragged_extra_vectors<FP, VK>() * VK;
  P.nK = (uint32_t)((P.Ks + tileK - 1) / tileK);
  if constexpr (RANK == 3) {
    // A last row tile that would hold only a few rows costs a whole column of workgroups (2^k+1 grids: ONE row in the 33rd
    // tile of 32, and 33 x 8 x 2 workgroups no longer fit the 512 slots that 32 x 8 x 2 fill exactly): those rows are left
    // to the caller's direct launch, next to the ragged row ends (MarchParams::rJ1 tells it which).  NEPTUNE_HIP_ROW_TAIL=0
    // keeps them here (measurements).
    static const bool row_tail = [] { const char* e = getenv("NEPTUNE_HIP_ROW_TAIL"); return !(e && e[0] == '0'); }();
    const int64_t rowsJ = P.rJ1 - P.rJ0, rem = rowsJ % tileJ;
    if (row_tail && rowsJ > tileJ && rem > 0 && rem * 8 <= tileJ) P.rJ1 -= (int32_t)rem;
  }
  P.nJ = (uint32_t)((P.rJ1 - P.rJ0 + tileJ - 1) / tileJ);
  const int64_t tilesJK = (int64_t)P.nJ * P.nK;
  int64_t chunk = chunk_req;
  if (chunk <= 0) {
    // Long chunks win -- every chunk re-reads 2*R0 planes and restarts the prefetch pipeline (measured:
    // profiles/r01_sweep_report.txt, profiles/r01_slab_chunks.txt; 128 planes = one chunk per 8-GPU slab
    // of the 1024^3 problem) -- but the grid runs in ROUNDS of (CUs x resident workgroups per CU)
    // workgroups, and a last round that is nearly empty costs as much as a full one (513^3 with
    // 128-plane chunks: 330 workgroups = 2 rounds on 256 CUs for the work of 1.3).  Pick the chunk
    // length that minimises rounds x (planes a workgroup streams, start-up included).
    static const int slots = [] {
      int dev = 0, cus = 256, per_cu = 1;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, march_kernel_fn<Body, T, RANK, NIN, FP, TL>(),
                                                       kWave * WJ * WK, 0) != hipSuccess || per_cu < 1)
        per_cu = 1;
      (void)hipGetLastError();
      return cus * per_cu;
    }();
    constexpr int R0 = (TL::JK2 && RANK == 2) ? 0 : FP::R0;
    // (up to 512 planes: 1024^3 on the 8-rows-per-lane tile is ONE round of 256 workgroups then, 2.75-2.78 ms against 2.79 with
    // 256-plane and 2.80 with 128-plane chunks; a 1024-plane chunk leaves half the CUs idle: 3.93 ms)
    const int cand3[] = {512, 256, 128, 96, 64, 48, 32, 24, 16}, cand2[] = {32, 24, 16, 8};
    const int* cand = RANK == 3 ? cand3 : cand2;
    const int ncand = RANK == 3 ? 9 : 4;
    int64_t best_cost = -1;
    auto consider = [&](int64_t len) {
      if (len > planes) len = planes;
      const int64_t blocks = ((planes + len - 1) / len) * tilesJK;
      const int64_t rounds = (blocks + slots - 1) / slots;
      const int64_t cost = rounds * (len + 2 * R0 + TL::PF + 2);
      if (best_cost < 0 || cost < best_cost) { best_cost = cost; chunk = len; }
    };
    for (int c = 0; c < ncand; ++c) consider(cand[c]);
    // ... and the lengths that FILL a whole number of rounds: as many chunks as r rounds hold, planes spread evenly over them
    // (1025 planes on 132 tiles and 256 slots: 13 chunks of 79 planes = 6.7 rounds of 7, against 11 chunks of 96 = 5.7 of 6)
    for (int64_t r = 1; r <= 64; ++r) {
      const int64_t n = r * slots / tilesJK;
      if (n < 1) continue;
      const int64_t len = (planes + n - 1) / n;
      if (len < cand[ncand - 1]) break;
      if (len <= cand[0]) consider(len);
    }
  }
  if (chunk > planes) chunk = planes;
  const int64_t nI = (planes + chunk - 1) / chunk;
  const int64_t blocks = nI * tilesJK;
  if (blocks <= 0 || blocks > 0x7fffffffLL) {
    fprintf(stderr, "[NeptuneRT][HIP] march grid of %lld workgroups is not launchable\n", (long long)blocks);
    abort();
  }
  P.chunk = (int32_t)chunk;
  const auto kern = march_kernel_fn<Body, T, RANK, NIN, FP, TL>();
  hipLaunchKernelGGL(kern, dim3((uint32_t)blocks),
                     dim3(kWave * WJ * WK), 0, stream, P, body);
  NEPTUNE_HIP_CHECK(hipGetLastError());
}

template <class Body, class T, int RANK, int NIN, class FP>
inline void launch_march(int variant, MarchParams<T, NIN>& P, const Body& body, int64_t planes, int chunk,
                         hipStream_t stream) {
  // footprints with wide register state (boxes, radius 2+, several halo inputs) cap the rows per lane at 4: on an
  // 8-row tile they would only spill (kMarchRowCap)
#define NEPTUNE_MV_CASE(idx, RJ, WJ, WK, DPP, NT, PF, NTL, LDSJ, JK, JHL, KD, PLN, name)                        \
  case idx:                                                                                                     \
    launch_march_variant<Body, T, RANK, NIN, FP,                                                                \
                         typename TileFor<T, FP, RANK, RJ, WJ, WK, DPP, NT, PF, NTL, LDSJ, JK, JHL, KD, PLN>::type>( \
        P, body, planes, chunk, stream);                                                                        \
    break;
  if constexpr (RANK == 3) {
    switch (variant) { NEPTUNE_MARCH3_VARIANTS(NEPTUNE_MV_CASE) default: abort(); }
  } else if constexpr (RANK == 1) {
    launch_march_variant<Body, T, RANK, NIN, FP, Tile<1, 1, 4, true, true, 1, false, false, false, false>>(P, body, planes,
                                                                                                        chunk, stream);
  } else {
    switch (variant) { NEPTUNE_MARCH2_VARIANTS(NEPTUNE_MV_CASE) default: abort(); }
  }
#undef NEPTUNE_MV_CASE
}

// ---- plan + launch ----------------------------------------------------------------------
template <class T, int RANK, int NIN, class FP>
inline int plan_apply(const neptune_hip_apply_geom_t* g, const void* const* in, const void* out,
                      const neptune_hip_launch_cfg_t* cfg) {
  constexpr int VK = 16 / sizeof(T);
  int rc = geom_validate(g);
  if (rc != NEPTUNE_HIP_OK) return rc;
  if (g->rank != RANK || g->num_inputs != NIN) return NEPTUNE_HIP_EINVAL;
  if (!in || !out) return NEPTUNE_HIP_EINVAL;
  for (int k = 0; k < NIN; ++k)
    if (!in[k]) return NEPTUNE_HIP_EINVAL;
  const int want = cfg ? cfg->kernel : NEPTUNE_HIP_KERNEL_AUTO;
  if (want == NEPTUNE_HIP_KERNEL_DIRECT) return NEPTUNE_HIP_KERNEL_DIRECT;

  bool ok = FP::MARCH_OK;
  const int64_t nK = g->out_ub[RANK - 1] - g->out_lb[RANK - 1];
  // ragged rows (nK % VK != 0): the kernel stores all whole vectors but the last, the rest of each row
  // goes to a direct-kernel launch (launch_apply below)
  const int64_t nK_march = (nK % VK == 0) ? nK : nK / VK * VK - VK - ragged_extra_vectors<FP, VK>() * VK;
  ok = ok && nK_march >= VK;
  {
    // in-plane offsets are 32-bit in the march kernel: every extent and one plane's bytes
    // (everything but dim 0) must stay below 2^31
    int64_t plane_bytes = (int64_t)sizeof(T);
    for (int d = 0; d < RANK; ++d) {
      const int64_t n = g->out_ub[d] - g->out_lb[d];
      ok = ok && n < 0x7fffffffLL;
      if (d > 0 || RANK == 1) plane_bytes *= n;  // rank 1: the single row is the "plane"
    }
    ok = ok && plane_bytes < 0x7fffffffLL;
  }
  for (int k = 0; ok && k < NIN; ++k) {
    // input 0 is also the copy-through source, read at the result's physical index (DataflowLowering.cpp:283-287): same
    // box.  Inputs 1.. index through their own lower bounds (:382-410): any box that contains the result's.
    bool same = true;
    int64_t kplane = (int64_t)sizeof(T);
    for (int d = 0; d < RANK; ++d) {
      same = same && g->in_lb[k][d] == g->out_lb[d] && g->in_ub[k][d] == g->out_ub[d];
      ok = ok && g->in_lb[k][d] <= g->out_lb[d] && g->in_ub[k][d] >= g->out_ub[d];
      const int64_t n = g->in_ub[k][d] - g->in_lb[k][d];
      ok = ok && n < 0x7fffffffLL;
      if (d > 0 || RANK == 1) kplane *= n;
    }
    ok = ok && kplane < 0x7fffffffLL && (k > 0 || same);
    // (cells of such an input right of the result's rows -- a field on the K faces, ghost cells -- may be read by the
    // result's last cells: at the end of a wave's span or an LDS window they arrive as halo cells, clamped per element; in
    // the middle of one they are the next lane's vector, whose load is clamped to the row's last whole vector and rotated
    // into place when it is consumed: InView::fix_k / fix_d)
    // rows of an input in the result's box start on 16-byte boundaries with it; a box of its own puts them anywhere (the
    // loads are then unaligned 16-byte accesses, like ragged rows)
    ok = ok && ((uintptr_t)in[k] % (same ? 16 : sizeof(T)) == 0);
  }
  ok = ok && ((uintptr_t)out % 16 == 0);
  if constexpr (RANK == 2) {
    // the LDS tile kernel keeps in-plane offsets in 32 bits and has no march form to fall back on
    if (tile2_only<T, FP, RANK>())
      ok = ok && (g->out_ub[0] - g->out_lb[0]) * (g->out_ub[1] - g->out_lb[1]) * (int64_t)sizeof(T) < 0x7fffffffLL;
  }
  for (int d = (RANK == 1 ? 0 : 1); ok && d < RANK; ++d)  // rank 1 has no marched dim to restrict
    ok = ok && g->region_lb[d] == 0 && g->region_ub[d] == g->out_ub[d] - g->out_lb[d];
  if (want == NEPTUNE_HIP_KERNEL_MARCH) return ok ? NEPTUNE_HIP_KERNEL_MARCH : NEPTUNE_HIP_EUNSUPPORTED;
  // auto: the march kernel pays off once a row fills at least one wave
  if (ok && nK_march >= (int64_t)kWave * VK) return NEPTUNE_HIP_KERNEL_MARCH;
  return NEPTUNE_HIP_KERNEL_DIRECT;
}

// geometry struct -> the direct kernel's parameter block (kernel axis order)
template <class T, int RANK, int NIN>
inline void fill_direct_params(const neptune_hip_apply_geom_t* g, const void* const* in, void* out, DirectParams<T, NIN>& P) {
  for (int k = 0; k < NIN; ++k) P.in[k] = static_cast<const T*>(in[k]);
  P.out = static_cast<T*>(out);
  int64_t shape[3];
  for (int d = 0; d < RANK; ++d) shape[d] = g->out_ub[d] - g->out_lb[d];
  to_axes<RANK>(shape, P.n, 1);
  to_axes<RANK>(g->out_lb, P.olb, 0);
  to_axes<RANK>(g->lb, P.lb, 0);
  to_axes<RANK>(g->ub, P.ub, 1);
  to_axes<RANK>(g->region_lb, P.rlb, 0);
  to_axes<RANK>(g->region_ub, P.rub, 1);
  for (int k = 0; k < NIN; ++k) {
    int64_t m[3], sh[3];
    for (int d = 0; d < RANK; ++d) {
      m[d] = g->in_ub[k][d] - g->in_lb[k][d];
      sh[d] = g->out_lb[d] - g->in_lb[k][d];
    }
    to_axes<RANK>(m, P.m[k], 1);
    to_axes<RANK>(sh, P.sh[k], 0);
  }
}

// The march tile a launch uses: cfg->variant if valid, else the automatic choice -- box stencils of rank 3
// take their own default tile, radius-2 stars or several halo inputs (more live rows per lane) the 2-row
// tile; in 2-D several halo inputs take the march form, whose state is one row per plane (the 4-row
// tile form spills from three halo inputs on; tools/multihalo_bench.sh, profiles/r01_multihalo.txt).
template <class T, int RANK, class FP>
inline int pick_march_variant(const neptune_hip_apply_geom_t* g, const neptune_hip_launch_cfg_t* cfg) {
  int variant = cfg ? cfg->variant : -1;
  constexpr int kNH = popcount_u(FP::HALO_MASK);
  constexpr bool kWideState = FP::R0 > 1 || FP::R1 > 1 || kNH > 1;
  if (variant < 0 || variant >= march_variant_count(RANK)) {
    // (2-D stars of radius 3-4 march down the rows like many-input applies: 8192^2 radius 4 measured 4.55 TB/s
    // against 3.2 on the tile form, whose row halo then is as tall as the tile; likewise the 25-point 5x5 box: 3.3
    // against 2.4)
    // 3-D stars of radius 2 and more (13-, 19-, 25-point operators ...): the plane-in-LDS kernel, four own rows per lane
    // (512^3 fp64: radius 2 4.96 TB/s against 4.53 on tile 2, radius 3 4.6 against 3.6, radius 4 4.1 against 2.8 on tile 5;
    // profiles/r02_plane.txt)
    // ... and from two inputs read at offsets on at any radius (two 7-point stars 512^3: 5.46 TB/s against 5.18 on tile 2)
    constexpr bool kPlane = plane_capable<FP, RANK>() && (FP::R0 > 1 || FP::R1 > 1 || FP::R2 > 1 || kNH > 1);
    variant = RANK == 3 ? (kPlane ? 7 : kWideState ? (FP::R0 > 3 ? 5 : 2) : FP::BOX ? 1 : 0)
                        : (RANK == 2 && (kNH > 2 || FP::R0 > 2 || (FP::BOX && FP::R0 > 1))) ? 1 : (RANK == 2 && kNH == 2) ? 2 : 0;
    if constexpr (RANK == 3) {
      // rows that fill the two-wave-wide tile badly (320 or 384 f64 cells against 256-cell tiles: a quarter to a
      // third of the lanes idle) take the one-wave-wide tile with twice the rows (measured +10 % at 320^3-640^3)
      if (variant == 0) {
        const int64_t n1 = g->out_ub[1] - g->out_lb[1], n2 = g->out_ub[2] - g->out_lb[2], span = kWave * (16 / (int64_t)sizeof(T));
        const int64_t wide = (n2 + 2 * span - 1) / (2 * span) * 2 * span, narrow = (n2 + span - 1) / span * span;
        if (wide * 20 > narrow * 21) variant = 1;
        // large fields: the 64-row tile, unless its rows-per-tile rounding wastes more than the 16-row tile's
        const int64_t tall = (n1 + 63) / 64 * 64, low = (n1 + 15) / 16 * 16;
        if (n1 >= 256 && tall * 100 <= low * 106) variant = 4;
        // the largest fields: the 8-rows-per-lane tile (32 rows x two wave spans), once even 128-plane chunks give every
        // CU two workgroups and more (1024^3: 1024 of them; 512^3 and the 8-GPU slabs of 1024^3 stay on tile 4, which
        // is as fast there and splits into more workgroups)
        if (variant == 4 || variant == 0) {
          const int64_t p0 = g->region_ub[0] - g->region_lb[0];
          const int64_t rows32 = (n1 + 31) / 32 * 32;
          if (rows32 * 100 <= n1 * 104 && ((n1 + 31) / 32) * ((n2 + 2 * span - 1) / (2 * span)) * ((p0 + 127) / 128) >= 512) variant = 6;
        }
      }
      // small fields: if even 16-plane chunks of the default tile give fewer workgroups than CUs (box stencils:
      // than 4 per CU -- their tile is register-heavy and gains from more, smaller workgroups up to ~400^3),
      // take the small tile (profiles/r01_size_sweep.txt)
      const MarchVariant* mv = march_variant(3, variant);
      const int64_t n1 = g->out_ub[1] - g->out_lb[1], n2 = g->out_ub[2] - g->out_lb[2];
      const int64_t p0 = g->region_ub[0] - g->region_lb[0];
      const int64_t tj = (int64_t)mv->RJ * mv->WJ, tk = (int64_t)mv->WK * kWave * (16 / (int64_t)sizeof(T));
      if (((n1 + tj - 1) / tj) * ((n2 + tk - 1) / tk) * ((p0 + 15) / 16) < (FP::BOX ? 1024 : 256)) variant = 3;
    }
  }
  if constexpr (RANK == 2) {
    // the rank-2 tile form treats the field as ONE plane and keeps in-plane offsets in 32 bits: a field of
    // 2 GiB and more takes the march form (row-restricted launches -- slab interiors and edges -- stay on the
    // tile form: MarchParams::rJ0/rJ1)
    const int64_t field_bytes = (g->out_ub[0] - g->out_lb[0]) * (g->out_ub[1] - g->out_lb[1]) * (int64_t)sizeof(T);
    if (march_variant(2, variant)->jk && field_bytes >= 0x7fffffffLL) variant = 1;
  }
  return variant;
}

// scratch bytes per lane (register spills) of one rank-3 march tile's kernel for this body; -1 if unknown.  The
// 8-rows-per-lane tile keeps ~200 VGPRs live for a 7-point body; a body with many more temporaries would spill
// there and is better off on the 4-row tile.
template <class Body, class T, int RANK, int NIN, class FP>
inline int march3_variant_scratch(int variant) {
  if constexpr (RANK != 3 || !FP::MARCH_OK) {
    return -1;
  } else {
    static std::mutex mu;
    static std::map<int, int> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(variant);
    if (it != cache.end()) return it->second;
    const void* fn = nullptr;
#define NEPTUNE_MV_FN(idx, RJ, WJ, WK, DPP, NT, PF, NTL, LDSJ, JK, JHL, KD, PLN, name) \
  case idx: fn = (const void*)march_kernel_fn<Body, T, RANK, NIN, FP, typename TileFor<T, FP, RANK, RJ, WJ, WK, DPP, NT, PF, NTL, LDSJ, false, JHL, KD, PLN>::type>(); break;
    switch (variant) { NEPTUNE_MARCH3_VARIANTS(NEPTUNE_MV_FN) default: break; }
#undef NEPTUNE_MV_FN
    int bytes = -1;
    hipFuncAttributes attr;
    if (fn && hipFuncGetAttributes(&attr, fn) == hipSuccess) bytes = (int)attr.localSizeBytes;
    else (void)hipGetLastError();
    cache[variant] = bytes;
    return bytes;
  }
}

// the direct kernel on g's region: rows form when all coordinates fit 31 bits (see apply_direct.hpp),
// else -- or when `flat` asks for it -- the flat form
template <class Body, class T, int RANK, int NIN>
inline int launch_direct(const Body& body, const neptune_hip_apply_geom_t* g, const void* const* in, void* out,
                         hipStream_t stream, bool flat) {
  DirectParams<T, NIN> P{};
  fill_direct_params<T, RANK, NIN>(g, in, out, P);
  {
    const int64_t lim = 0x7fffff00LL;
    const int64_t eK = P.rub[2] - P.rlb[2], rows = (P.rub[0] - P.rlb[0]) * (P.rub[1] - P.rlb[1]);
    bool narrow = !flat && P.n[0] * P.n[1] < lim && P.n[2] < lim;
    for (int k = 0; k < NIN; ++k) {
      narrow = narrow && P.m[k][0] * P.m[k][1] < lim && P.m[k][2] < lim;
      for (int ax = 0; ax < 3; ++ax) narrow = narrow && P.sh[k][ax] > -lim && P.sh[k][ax] < lim;
    }
    const int64_t nchunk = (eK + 255) / 256;
    if (narrow && rows * nchunk < lim) {
      hipLaunchKernelGGL((neptune_apply_rows<Body, T, RANK, NIN>), grid_for_blocks(rows * nchunk), dim3(256), 0, stream, P,
                         body, (uint32_t)nchunk);
      NEPTUNE_HIP_CHECK(hipGetLastError());
      return NEPTUNE_HIP_OK;
    }
  }
  const int64_t total = (P.rub[0] - P.rlb[0]) * (P.rub[1] - P.rlb[1]) * (P.rub[2] - P.rlb[2]);
  const int64_t blocks = (total + 255) / 256;
  if (blocks > 0x7fffffffLL) {
    fprintf(stderr, "[NeptuneRT][HIP] direct grid of %lld workgroups is not launchable\n", (long long)blocks);
    abort();
  }
  hipLaunchKernelGGL((neptune_apply_direct<Body, T, RANK, NIN>), grid_for_blocks(blocks), dim3(256), 0, stream, P,
                     body);
  NEPTUNE_HIP_CHECK(hipGetLastError());
  return NEPTUNE_HIP_OK;
}

template <class Body, class T, int RANK, int NIN, class FP>
inline int launch_apply_impl(const Body& body, const neptune_hip_apply_geom_t* g, const void* const* in, void* out,
                             hipStream_t stream, const neptune_hip_launch_cfg_t* cfg) {
  const int kernel = plan_apply<T, RANK, NIN, FP>(g, in, out, cfg);
  if (kernel < 0) return kernel;
  for (int d = 0; d < RANK; ++d)
    if (g->region_lb[d] == g->region_ub[d]) return NEPTUNE_HIP_OK;  // empty region: nothing to do

  if constexpr (FP::MARCH_OK) if (kernel == NEPTUNE_HIP_KERNEL_MARCH) {
    MarchParams<T, NIN> P{};
    for (int k = 0; k < NIN; ++k) P.in[k] = static_cast<const T*>(in[k]);
    P.out = static_cast<T*>(out);
    int variant = pick_march_variant<T, RANK, FP>(g, cfg);
    if (RANK == 3 && variant == 6 && !(cfg && cfg->variant == 6) && march3_variant_scratch<Body, T, RANK, NIN, FP>(6) > 0)
      variant = 4;  // automatic choice only: this body spills on the 8-rows-per-lane tile
    // rank-2 tile form: (d0,d1) -> (J,K), one plane (the LDS tile kernel is a tile form whatever the table row says)
    const bool jk = RANK == 2 && (march_variant(RANK, variant)->jk || tile2_only<T, FP, RANK>());
    auto axes = [&](const int64_t* src, int64_t dst[3], int64_t fill) {
      if (jk) { dst[0] = fill; dst[1] = src[0]; dst[2] = src[1]; }
      else to_axes<RANK>(src, dst, fill);
    };
    int64_t shape[3], plb[3], pub[3], n[3], a_lb[3], a_ub[3], rlb[3], rub[3];
    for (int d = 0; d < RANK; ++d) {
      shape[d] = g->out_ub[d] - g->out_lb[d];
      plb[d] = g->lb[d] - g->out_lb[d];  // result-physical bounds
      pub[d] = g->ub[d] - g->out_lb[d];
    }
    axes(shape, n, 1);
    axes(plb, a_lb, 0);
    axes(pub, a_ub, 1);
    P.N0 = (int32_t)n[0]; P.N1 = (int32_t)n[1]; P.N2 = (int32_t)n[2];
    for (int a = 0; a < 3; ++a) {
      // clamp into [0, extent] so the int32 narrowing is exact (empty bounds stay empty)
      const int64_t lo = a_lb[a] < 0 ? 0 : (a_lb[a] > n[a] ? n[a] : a_lb[a]);
      const int64_t hi = a_ub[a] < 0 ? 0 : (a_ub[a] > n[a] ? n[a] : a_ub[a]);
      P.plb[a] = (int32_t)lo;
      P.pub[a] = (int32_t)hi;
    }
    axes(g->out_lb, P.olb, 0);
    if constexpr (NIN > 1) {
      // per-input views (MarchParams::view): pitch, shift and the result-physical range each input holds, per kernel axis
      for (int k = 0; k < NIN; ++k) {
        int64_t kshape[3], ksh[3], kn[3], s3[3];
        for (int d = 0; d < RANK; ++d) {
          kshape[d] = g->in_ub[k][d] - g->in_lb[k][d];
          ksh[d] = g->out_lb[d] - g->in_lb[k][d];
        }
        axes(kshape, kn, 1);
        axes(ksh, s3, 0);
        InView& vw = P.view[k];
        vw.row_b = (int32_t)(kn[2] * (int64_t)sizeof(T));
        vw.plane_b = kn[1] * kn[2] * (int64_t)sizeof(T);
        for (int a = 0; a < 3; ++a) {
          vw.sh[a] = (int32_t)s3[a];
          vw.lo[a] = (int32_t)-s3[a];
          vw.hi[a] = (int32_t)(kn[a] - s3[a] - 1);
        }
        // the first lane vector (starts are multiples of VK) that does not fit the input's row as a whole, if the row still
        // holds cells of it
        constexpr int VKc = 16 / (int)sizeof(T);
        const int32_t kmax = vw.hi[2] - VKc + 1;
        const int32_t kstar = kmax < 0 ? 0 : (kmax / VKc + 1) * VKc;     // smallest multiple of VK above kmax
        vw.fix_k = kstar;
        vw.fix_d = (kstar <= vw.hi[2] && kstar - kmax < VKc) ? kstar - kmax : 0;
      }
    }
    P.rJ0 = 0; P.rJ1 = P.N1;
    if (jk) {  // one plane; a launch region restricted along d0 is a row range of it
      P.rI0 = 0; P.rI1 = 1;
      P.rJ0 = (int32_t)g->region_lb[0]; P.rJ1 = (int32_t)g->region_ub[0];
    } else {
      to_axes<RANK>(g->region_lb, rlb, 0);
      to_axes<RANK>(g->region_ub, rub, 1);
      P.rI0 = (int32_t)rlb[0]; P.rI1 = (int32_t)rub[0];
    }
    launch_march<Body, T, RANK, NIN, FP>(variant, P, body, P.rI1 - P.rI0, cfg ? cfg->chunk : 0, stream);
    neptune_hip_note_launch(NEPTUNE_HIP_KERNEL_MARCH, variant, P.chunk);
    int rc = NEPTUNE_HIP_OK;
    if (P.N2 % (16 / (int)sizeof(T)) != 0) {
      // ragged rows: cells [Ks, N2) of every row the march launch stored -- fewer than 3*VK per row -- through the
      // flat direct kernel (lanes run down the rows: strided, but a fraction of a percent of the field)
      neptune_hip_apply_geom_t tail = *g;
      tail.region_lb[RANK - 1] = P.Ks;
      if (RANK == 3 && !jk) tail.region_ub[1] = P.rJ1;
      rc = launch_direct<Body, T, RANK, NIN>(body, &tail, in, out, stream, true);
    }
    if (RANK == 3 && !jk && P.rJ1 < P.N1 && rc == NEPTUNE_HIP_OK) {
      // the few rows past the last whole row tile (launch_march_variant): whole rows, lanes along them
      neptune_hip_apply_geom_t rows = *g;
      rows.region_lb[1] = P.rJ1;
      rc = launch_direct<Body, T, RANK, NIN>(body, &rows, in, out, stream, false);
    }
    return rc;
  }

  neptune_hip_note_launch(NEPTUNE_HIP_KERNEL_DIRECT, -1, 0);
  return launch_direct<Body, T, RANK, NIN>(body, g, in, out, stream, cfg && (cfg->flags & NEPTUNE_HIP_FLAG_DIRECT_FLAT));
}

// ---- measured launch choice at first use, remembered as wisdom -----------------------------------------------
// FFTW_MEASURE for applies, on by default for fields large enough to measure: the first launch of a (body, geometry)
// with no explicit configuration looks the choice up in the wisdom file (include/neptune_hip.h "launch wisdom"); if it
// is not there it times the default march tiles this translation unit holds with a few chunk lengths, keeps the fastest
// for later launches of the same geometry and appends it to the file, so that every later process starts from it without
// timing anything.  Every candidate computes the same bits into `out`, so the timed launches are harmless; they
// synchronise the stream, which is why a launch inside a stream capture never measures.
//   NEPTUNE_HIP_TUNE=0  no measuring, fixed automatic tiles      NEPTUNE_HIP_TUNE=1  measure fields of any size
//   NEPTUNE_HIP_TUNE_MIN_CELLS  smallest launch region that is measured by default (2^24 cells)
#ifndef NEPTUNE_HIP_MODULE_ID
#define NEPTUNE_HIP_MODULE_ID "lib"   // emitted modules define it: a hash of their body functors
#endif
#ifndef NEPTUNE_HIP_BUILD_ID
#define NEPTUNE_HIP_BUILD_ID "dev"    // the build defines it: a hash of the kernel headers this code was compiled from
#endif
inline int tune_mode() {   // 0 = never, 1 = large launches (default), 2 = every launch
  static const int mode = [] {
    const char* e = getenv("NEPTUNE_HIP_TUNE");
    if (!e || !*e) return 1;
    return *e == '0' ? 0 : 2;
  }();
  return mode;
}
inline int64_t tune_min_cells() {
  static const int64_t n = [] {
    const char* e = getenv("NEPTUNE_HIP_TUNE_MIN_CELLS");
    const long long v = (e && *e) ? atoll(e) : 0;
    return (int64_t)(v > 0 ? v : (1LL << 24));
  }();
  return n;
}
inline bool tune_enabled() { return tune_mode() == 2; }
// how many tiles of each rank's table are the defaults (what a lowered module holds)
#define NEPTUNE_MV_ONE(idx, RJ, WJ, WK, DPP, NT, PF, NTL, LDSJ, JK, JHL, KD, PLN, name) +1
constexpr int kNumMarch3Default = 0 NEPTUNE_MARCH3_DEFAULT(NEPTUNE_MV_ONE);
constexpr int kNumMarch2Default = 0 NEPTUNE_MARCH2_DEFAULT(NEPTUNE_MV_ONE);
#undef NEPTUNE_MV_ONE

template <class Body, class T, int RANK, int NIN, class FP>
inline neptune_hip_launch_cfg_t tune_apply(const Body& body, const neptune_hip_apply_geom_t* g, const void* const* in, void* out,
                                           hipStream_t stream, double* best_ms_out = nullptr) {
  neptune_hip_launch_cfg_t best = {NEPTUNE_HIP_KERNEL_AUTO, -1, 0, 0};
  if (plan_apply<T, RANK, NIN, FP>(g, in, out, &best) != NEPTUNE_HIP_KERNEL_MARCH) return best;
  hipEvent_t e0, e1, e2;
  NEPTUNE_HIP_CHECK(hipEventCreate(&e0));
  NEPTUNE_HIP_CHECK(hipEventCreate(&e1));
  NEPTUNE_HIP_CHECK(hipEventCreate(&e2));
  float best_ms = -1.f;
  // one launch alone first (it also pays one-off costs); a candidate whose first launch already takes 1.4x the best
  // average so far is dropped without its repetitions (the spilling tiles of a fat body: 2-6x)
  auto time_cfg = [&](const neptune_hip_launch_cfg_t& c) -> float {
    NEPTUNE_HIP_CHECK(hipEventRecord(e0, stream));
    if (launch_apply_impl<Body, T, RANK, NIN, FP>(body, g, in, out, stream, &c) != NEPTUNE_HIP_OK) return -1.f;
    NEPTUNE_HIP_CHECK(hipEventRecord(e1, stream));
    NEPTUNE_HIP_CHECK(hipEventSynchronize(e1));
    float first = 0, ms = 0;
    NEPTUNE_HIP_CHECK(hipEventElapsedTime(&first, e0, e1));
    if (best_ms > 0 && first > 1.4f * best_ms) return first;
    // repetitions for about 4 ms of launches, 3 to 20: the tiles of a 0.2-0.4 ms launch lie within 2-3 % of each other, which
    // three repetitions do not resolve (the same geometry then gets different tiles on different boxes)
    const int reps = first > 0.f ? (int)fminf(20.f, fmaxf(3.f, ceilf(4.f / first))) : 3;
    for (int r = 0; r < reps; ++r) launch_apply_impl<Body, T, RANK, NIN, FP>(body, g, in, out, stream, &c);
    NEPTUNE_HIP_CHECK(hipEventRecord(e2, stream));
    NEPTUNE_HIP_CHECK(hipEventSynchronize(e2));
    NEPTUNE_HIP_CHECK(hipEventElapsedTime(&ms, e1, e2));
    return ms / (float)reps;
  };
  auto try_cfg = [&](const neptune_hip_launch_cfg_t& c) {
    const float ms = time_cfg(c);
    if (ms > 0 && (best_ms <= 0 || ms < best_ms)) { best_ms = ms; best = c; }
    return ms;
  };
  try_cfg(best);
  try_cfg(best);   // the very first launches of a process run on clocks that are still ramping: time the automatic choice twice
  // stage 1: every default tile on its automatic chunk length; stage 2: chunk lengths on the two fastest tiles
  // (a module built with NEPTUNE_HIP_FULL_VARIANTS=1 asked for every tile; the runtime library holds them all for the
  // sweeps and parity tests, its first-use choice stays among the defaults)
  constexpr int kDefaults = RANK == 3 ? kNumMarch3Default : (RANK == 2 ? kNumMarch2Default : 1);
  const bool all_tiles = NEPTUNE_HIP_FULL_VARIANTS && strcmp(NEPTUNE_HIP_MODULE_ID, "lib") != 0;
  const int nv = (all_tiles || march_variant_count(RANK) < kDefaults) ? march_variant_count(RANK) : kDefaults;
  int top[2] = {-1, -1};
  float top_ms[2] = {-1.f, -1.f};
  for (int v = 0; v < nv; ++v) {
    const float ms = try_cfg({NEPTUNE_HIP_KERNEL_MARCH, v, 0, 0});
    if (ms <= 0) continue;
    if (top_ms[0] < 0 || ms < top_ms[0]) { top[1] = top[0]; top_ms[1] = top_ms[0]; top[0] = v; top_ms[0] = ms; }
    else if (top_ms[1] < 0 || ms < top_ms[1]) { top[1] = v; top_ms[1] = ms; }
  }
  const int64_t planes = RANK == 1 ? 1 : g->region_ub[0] - g->region_lb[0];
  const int chunks3[] = {32, 64, 128, 256, 512}, chunks2[] = {32};
  for (int t = 0; t < 2; ++t) {
    if (top[t] < 0) continue;
    const bool tile2 = RANK == 2 && march_variant(RANK, top[t])->jk;
    if (RANK == 1 || tile2) continue;   // one plane: no chunk length to choose
    const int* chunks = RANK == 3 ? chunks3 : chunks2;
    const int nc = RANK == 3 ? 5 : 1;
    for (int c = 0; c < nc; ++c)
      if (chunks[c] < planes) try_cfg({NEPTUNE_HIP_KERNEL_MARCH, top[t], chunks[c], 0});
  }
  // play-off: the winner against the automatic choice and the fastest tile's automatic chunk, each measured again in turn
  // (the first candidates of a process are timed on clocks that are still ramping; a choice that only looked faster must
  // not be written into the wisdom file)
  {
    neptune_hip_launch_cfg_t finals[3] = {best, {NEPTUNE_HIP_KERNEL_AUTO, -1, 0, 0}, {NEPTUNE_HIP_KERNEL_MARCH, top[0], 0, 0}};
    const int nf = top[0] >= 0 ? 3 : 2;
    float fin_ms[3] = {-1.f, -1.f, -1.f};
    best_ms = -1.f;   // time_cfg's early exit compares with it: measure all finalists in full
    for (int round = 0; round < 2; ++round)
      for (int f = 0; f < nf; ++f) {
        const float a = time_cfg(finals[f]), b = time_cfg(finals[f]);
        const float ms = (a > 0 && b > 0) ? (a < b ? a : b) : (a > 0 ? a : b);
        if (ms > 0 && (fin_ms[f] < 0 || ms < fin_ms[f])) fin_ms[f] = ms;
      }
    int win = 0;
    for (int f = 1; f < nf; ++f)
      if (fin_ms[f] > 0 && (fin_ms[win] <= 0 || fin_ms[f] < fin_ms[win])) win = f;
    best = finals[win];
    best_ms = fin_ms[win];
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipEventDestroy(e2);
  if (best_ms_out) *best_ms_out = best_ms;
  return best;
}

template <class Body, class T, int RANK, int NIN, class FP>
inline int launch_apply(const Body& body, const neptune_hip_apply_geom_t* g, const void* const* in, void* out,
                        hipStream_t stream, const neptune_hip_launch_cfg_t* cfg) {
  const bool free_choice = !cfg || (cfg->kernel == NEPTUNE_HIP_KERNEL_AUTO && cfg->variant < 0 && cfg->chunk == 0 && cfg->flags == 0);
  if (free_choice && tune_mode() != 0 && g && FP::MARCH_OK) {
    // one table per body (this function is instantiated per Body); key: everything of the geometry that the
    // launcher looks at, plus the 16-byte alignment of the buffers
    static std::mutex mu;
    static std::map<std::array<int64_t, 20>, neptune_hip_launch_cfg_t> table;
    std::array<int64_t, 20> key{};
    int n = 0;
    int64_t cells = 1;
    for (int d = 0; d < 3; ++d) {
      key[n++] = d < RANK ? g->out_ub[d] - g->out_lb[d] : 1;
      key[n++] = d < RANK ? g->lb[d] - g->out_lb[d] : 0;
      key[n++] = d < RANK ? g->ub[d] - g->out_lb[d] : 1;
      key[n++] = d < RANK ? g->region_lb[d] : 0;
      key[n++] = d < RANK ? g->region_ub[d] : 1;
      if (d < RANK) cells *= (g->region_ub[d] > g->region_lb[d] ? g->region_ub[d] - g->region_lb[d] : 0);
    }
    int64_t align = ((uintptr_t)out % 16 == 0);
    for (int k = 0; k < NIN; ++k) {
      align = align * 2 + ((uintptr_t)in[k] % 16 == 0);
      for (int d = 0; d < RANK; ++d) align = align * 2 + (g->in_lb[k][d] == g->out_lb[d] && g->in_ub[k][d] == g->out_ub[d]);
    }
    key[n++] = align;
    const neptune_hip_launch_cfg_t untuned = {NEPTUNE_HIP_KERNEL_AUTO, -1, 0, 0};
    neptune_hip_launch_cfg_t tuned = untuned;
    bool have = false;
    {
      std::lock_guard<std::mutex> lk(mu);
      auto it = table.find(key);
      if (it != table.end()) { tuned = it->second; have = true; }
    }
    if (!have) {
      bool decided = true;   // false: leave the table alone (a capture is in progress: measure at the next plain launch)
      if ((tune_mode() == 2 || cells >= tune_min_cells()) && geom_validate(g) == NEPTUNE_HIP_OK &&
          plan_apply<T, RANK, NIN, FP>(g, in, out, &untuned) == NEPTUNE_HIP_KERNEL_MARCH) {
        // wisdom key: kernel build, module, body type, element size, rank, inputs, then the table key
        std::string wkey = std::string(NEPTUNE_HIP_BUILD_ID) + "|" + NEPTUNE_HIP_MODULE_ID + "|" + typeid(Body).name() + "|" +
                           std::to_string(sizeof(T)) + "|" + std::to_string(RANK) + "|" + std::to_string(NIN);
        for (int i = 0; i < n; ++i) wkey += (i ? "," : "|") + std::to_string(key[i]);
        if (neptune_hip_wisdom_lookup(wkey.c_str(), &tuned) &&
            (tuned.kernel != NEPTUNE_HIP_KERNEL_MARCH || (tuned.variant >= 0 && tuned.variant < march_variant_count(RANK)))) {
          // (a choice this translation unit cannot run -- a tile index beyond its table -- is measured again)
        } else {
          tuned = untuned;
          // measuring synchronises the stream: not possible while it is being captured into a graph (the step
          // loop launches once outside capture first, so its graph still gets the measured choice)
          hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
          if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
          if (cs == hipStreamCaptureStatusNone) {
            double ms = 0;
            tuned = tune_apply<Body, T, RANK, NIN, FP>(body, g, in, out, stream, &ms);
            (void)neptune_hip_wisdom_store(wkey.c_str(), &tuned, ms);
          } else {
            decided = false;
          }
        }
      }
      if (decided) {
        std::lock_guard<std::mutex> lk(mu);
        table[key] = tuned;
      }
    }
    const bool is_tuned = tuned.kernel != NEPTUNE_HIP_KERNEL_AUTO || tuned.variant >= 0 || tuned.chunk != 0;
    if (is_tuned) return launch_apply_impl<Body, T, RANK, NIN, FP>(body, g, in, out, stream, &tuned);
  }
  return launch_apply_impl<Body, T, RANK, NIN, FP>(body, g, in, out, stream, cfg);
}

}  // namespace neptune_hip

// apply_march2.hpp -- TWO chained applies of the same body in ONE pass over HBM (temporal blocking):
//     out = A(A(in))     with A = one neptune_ir.apply (copy-through outside its bounds included)
//
// An explicit time loop `u <- A(u)` repeated is bound by HBM: every step reads the field once and writes it
// once.  The reference runs such loops on the host, one apply per step (runtime method 2: forward Euler,
// lib/Runtime/PETSc/NeptunePETScRuntime.cpp:677-712).  This kernel produces step n+2 from step n directly: the
// intermediate field v = A(u) exists only in registers, so two steps cost one read and one write of the field plus
// the redundant rim described below -- 2.3 field passes instead of 4.1.  The operations, their order and their operands
// are exactly those of two separate launches (the body functor is evaluated once per cell and stage, strict IEEE,
// no contraction), hence the same bits.
//
// Scope: rank 3: star footprints of input 0 up to radius 2 per axis (the 7-point family, the fused explicit Euler step of
// such an operator included; 13-point 4th-order operators: two applies per pass), further inputs read at the centre only
// (coefficient fields: the same field at every stage); rank 2: the 5-point family.  All boxes equal, rows a whole number of
// 64-byte store granules.  Everything else keeps using one launch per apply (neptune_hip_step_loop decides).
//
// Shape of the march (one workgroup = WJ waves stacked along J, one wave span wide):
//   * a wave owns RJ rows x 64 lane vectors; the workgroup's window is TJ = WJ*RJ rows x 64*VK cells.
//   * stage 1 computes v on the whole window from u (K neighbours by wave shifts, J neighbours of the window's own
//     rows through LDS); v is therefore valid one cell / one row inside the window's edge, and stage 2's result w two
//     cells / rows inside.  Windows OVERLAP instead of fetching halos: the row stride between workgroups is TJ - 4 and
//     the column stride one 64-byte granule less than the wave span (120 of 128 fp64 cells), with the kept columns
//     starting half a granule inside the window: every store instruction then writes whole 64-byte granules (a
//     partially written granule costs a read-modify-write at the memory side: profiles/r02_ragged_probe.txt).  No scalar
//     halo loads at all, no workgroup-edge row loads.
//   * along dim 0 the wave keeps 3 planes of u and 3 planes of v in registers; step i loads u(i+3), computes v(i+1)
//     from u(i..i+2) and w(i) from v(i-1..i+1), stores w(i).  A chunk starts two planes early (results discarded)
//     to fill the v ring: 4 redundant plane reads per chunk.
//   * one barrier per step: the waves publish the edge rows of u(i+1) AND of v(i) together.
#pragma once
#include "apply_march.hpp"

namespace neptune_hip {

template <class T, int NIN>
struct March2Params {
  const T* in[NIN];        // in[0]: the field the applies chain on; in[1..]: inputs read at the centre only, the same at every stage
  T* out;
  int32_t N0, N1, N2;
  int32_t plb[3], pub[3];  // apply.bounds, result-physical
  int64_t olb[3];          // logical origin (index arguments)
  int32_t rI0, rI1;        // planes this launch stores
  int32_t chunk;
  uint32_t nJ, nK;
};

// window constants of a footprint: each stage loses R1 rows / R2 columns / R0 planes of validity per side
template <class T, class FP, int NS>
struct March2Geom {
  static constexpr int VK = 16 / (int)sizeof(T), G = 64 / (int)sizeof(T);   // cells per lane vector / per 64-byte store granule
  static constexpr int R0 = FP::R0, R1 = FP::R1, R2 = FP::R2;
  // kept columns start MK cells inside the window: a whole number of half granules (so that both ends of the kept span are
  // granule boundaries: SPAN - 2 MK is a multiple of G) and of lane vectors, at least NS * R2
  static constexpr int HG = G / 2 > VK ? G / 2 : VK;
  static constexpr int MK = (NS * R2 + HG - 1) / HG * HG;
  static constexpr int SPAN = kWave * VK, KEEPK = SPAN - 2 * MK;
  static constexpr int MJ = NS * R1;                                        // rows lost per side
  static constexpr int WARM = 2 * R0 * (NS - 1);                            // planes a chunk starts early
};

// NS chained applies per pass (2 or 3); RJ rows per lane, WJ waves per workgroup, MINW = waves per SIMD the register
// allocation must leave room for.  Stage k = 1..NS computes v_k = A(v_{k-1}) (v_0 = the input u, v_NS = the result w);
// at step i stage k produces plane i + (NS - k) R0, so each stage's newest plane is the next stage's farthest upper
// neighbour plane within the same step.  v_k is valid k R cells / rows inside the window.
// Star footprints of input 0 up to radius 2 per axis (7-point family, 13-point 4th-order operators); inputs 1.. are read at
// the centre only and are the same field at every stage (coefficient fields): each keeps a queue of the (NS-1) R0 + 1 planes
// between the first stage's plane and the last one's.
template <class Body, class T, int NIN, class FP, int NS, int RJ, int WJ, int MINW>
__global__ __launch_bounds__(kWave* WJ, MINW) void neptune_apply_march2(March2Params<T, NIN> P, Body body) {
  using V = typename Vec16<T>::type;
  using GM = March2Geom<T, FP, NS>;
  constexpr int VK = GM::VK, R0 = GM::R0, R1 = GM::R1, R2 = GM::R2, NP = 2 * R0 + 1;
  static_assert(NS >= 2 && NS <= 3, "two or three applies per pass");
  constexpr int TJ = RJ * WJ;                 // window rows
  constexpr int MK = GM::MK, SPAN = GM::SPAN;
  static_assert(R0 >= 1 && R0 <= 2 && R1 >= 1 && R1 <= 2 && R2 >= 1 && R2 <= 2 && !FP::BOX && FP::HALO_MASK == 1u, "star of input 0, radius 1..2");
  static_assert(RJ >= R1 && R2 <= VK && TJ > 2 * GM::MJ, "window too small for the footprint");
  constexpr int NQ = (NS - 1) * R0 + 1;       // planes of a centre-only input between the first stage's plane and the last one's
  constexpr int NF = NIN > 1 ? NIN - 1 : 1;
  __shared__ V lds[2][WJ][NS][2 * R1][kWave];  // [parity][wave][stage input k][first R1 own rows | last R1 own rows][lane]

  const int lane = threadIdx.x & (kWave - 1);
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t vb = xcd_remap(blockIdx.x, gridDim.x);
  const uint32_t kt = vb % P.nK;
  const uint32_t t = vb / P.nK;
  const uint32_t jt = t % P.nJ;
  const uint32_t ct = t / P.nJ;

  const int32_t Jb = (int32_t)(jt * (TJ - 2 * GM::MJ)) - GM::MJ;   // first row of the window
  const int32_t j0 = Jb + w * RJ;                                    // first own row
  const int32_t kw = (int32_t)(kt * GM::KEEPK) - MK;                 // first column of the window
  const int32_t k0 = kw + lane * VK;
  // loads: clamped into the field (a clamped cell only ever feeds values that are discarded)
  const int32_t kc = k0 < 0 ? 0 : (k0 > P.N2 - VK ? P.N2 - VK : k0);
  const uint32_t lane_b = (uint32_t)kc * (uint32_t)sizeof(T);
  uint32_t rowb[RJ];
  static_for<RJ>([&](auto rc) {
    constexpr int r = rc;
    int32_t j = j0 + r;
    j = j < 0 ? 0 : (j >= P.N1 ? P.N1 - 1 : j);
    rowb[r] = (uint32_t)j * (uint32_t)P.N2 * (uint32_t)sizeof(T);
  });
  const int64_t plane_b = (int64_t)P.N1 * P.N2 * (int64_t)sizeof(T);

  const int32_t ib = P.rI0 + (int32_t)ct * P.chunk;
  const int32_t ie = (ib + P.chunk < P.rI1) ? ib + P.chunk : P.rI1;
  if (ib >= ie) return;

  auto load_plane = [&](const T* field, int32_t ip, V(&rows)[RJ]) {
    const int32_t ic = ip < 0 ? 0 : (ip >= P.N0 ? P.N0 - 1 : ip);
    const char* base = reinterpret_cast<const char*>(field) + (int64_t)ic * plane_b;
    static_for<RJ>([&](auto rc) {
      constexpr int r = rc;
      rows[r] = *reinterpret_cast<const V*>(base + (rowb[r] + lane_b));
    });
  };

  // store predicates: rows and columns of w this wave keeps
  bool row_keep[RJ], in_j[RJ], in_k[VK];
  static_for<RJ>([&](auto rc) {
    constexpr int r = rc;
    const int32_t j = j0 + r;
    row_keep[r] = j >= Jb + GM::MJ && j < Jb + TJ - GM::MJ && j >= 0 && j < P.N1;
    in_j[r] = j >= P.plb[1] && j < P.pub[1];
  });
  static_for<VK>([&](auto ec) {
    constexpr int e = ec;
    in_k[e] = (k0 + e) >= P.plb[2] && (k0 + e) < P.pub[2];
  });
  const bool lane_keep = k0 >= kw + MK && k0 < kw + SPAN - MK && k0 >= 0 && k0 < P.N2;

  // one stage: `pl` = the 2 R0 + 1 planes of the stage's input (own rows), centre at pl[R0]; `above` / `below` the R1 rows
  // outside the wave's own rows on the centre plane; `fx` the centre-only inputs' rows on the stage's plane
  auto stage = [&](const V(&pl)[NP][RJ], const V(&above)[R1], const V(&below)[R1], const V(&fx)[NF][NQ][RJ], auto qc, int32_t ip,
                   V(&res)[RJ]) {
    constexpr int q = decltype(qc)::value;
    V ring[1][NP][RJ + 2 * R1];
    T lft[1][1][RJ + 2 * R1][R2], rgt[1][1][RJ + 2 * R1][R2];
    V pt[NIN][RJ];
    static_for<RJ>([&](auto rc) {
      constexpr int r = rc;
      static_for<NP>([&](auto pc) { constexpr int pp = pc; ring[0][pp][r + R1] = pl[pp][r]; });
      // K neighbours from the adjacent lanes; the wave's outermost cells get an arbitrary edge value: their results
      // lie outside the kept columns
      static_for<R2>([&](auto xc) {
        constexpr int x = xc;
        constexpr int dl = R2 - x, dr = x + 1;
        lft[0][0][r + R1][x] = from_prev<true>(pl[R0][r][VK - dl], pl[R0][r][0], lane);
        rgt[0][0][r + R1][x] = from_next<true>(pl[R0][r][dr - 1], pl[R0][r][VK - 1], lane);
      });
      static_for<NIN>([&](auto nc) { constexpr int n = nc; if constexpr (n > 0) pt[n][r] = fx[n - 1][q][r]; });
    });
    static_for<R1>([&](auto xc) {
      constexpr int x = xc;
      ring[0][R0][x] = above[x];
      ring[0][R0][RJ + R1 + x] = below[x];
    });
    const bool in_i = ip >= P.plb[0] && ip < P.pub[0];
    const int64_t li = (int64_t)ip + P.olb[0];
    static_for<RJ>([&](auto rc) {
      constexpr int r = rc;
      const int64_t lj = (int64_t)(j0 + r) + P.olb[1];
      static_for<VK>([&](auto ec) {
        constexpr int e = ec;
        const int64_t lk = (int64_t)(k0 + e) + P.olb[2];
        MarchAcc<T, 3, NIN, FP, RJ, r, e, false> acc{ring, lft, rgt, pt, li, lj, lk};
        const T val = body(acc);
        const T through = pl[R0][r][e];
        res[r][e] = (in_i && in_j[r] && in_k[e]) ? val : OutsideOf<Body, T>::apply(body, through);
      });
    });
  };

  // ---- register state.  ring[k] holds the 2 R0 + 1 newest planes of v_k that stage k+1 reads: at step i planes
  // i + (NS-k) R0 - 2 R0 .. i + (NS-k) R0, the centre of stage k+1 in the middle; ring[k][2 R0] is written by stage k in the
  // same step (k = 0: by the load issued one step earlier).  fx[n][q]: plane i + q R0 ... of centre-only input n+1: slot q is
  // the plane of the stage that works q R0 planes ahead of the last one.
  V ring[NS][NP][RJ];
  V un[RJ];                                // u(i + NS R0 + 1) in flight
  V fx[NF][NQ][RJ], fn[NF][RJ];            // centre-only inputs: queue and the plane in flight
  V wres[RJ];
  const int32_t i0 = ib - GM::WARM;        // warm-up steps fill the rings of the intermediate fields (their w is discarded)
  static_for<NP - 1>([&](auto pc) { constexpr int pp = pc; load_plane(P.in[0], i0 + NS * R0 - 2 * R0 + pp, ring[0][pp]); });
  load_plane(P.in[0], i0 + NS * R0, un);
  if constexpr (NIN > 1) {
    static_for<NIN - 1>([&](auto nc) {
      constexpr int n = nc;
      static_for<NQ - 1>([&](auto qc) { constexpr int q = qc; load_plane(P.in[n + 1], i0 + q, fx[n][q]); });
      load_plane(P.in[n + 1], i0 + NQ - 1, fn[n]);
    });
  }
  static_for<NS - 1>([&](auto kc) {        // defined, never part of a kept result
    constexpr int k = kc + 1;
    static_for<NP - 1>([&](auto pc) { constexpr int pp = pc; static_for<RJ>([&](auto rc) { constexpr int r = rc; ring[k][pp][r] = ring[0][0][r]; }); });
  });

  for (int32_t i = i0; i < ie; ++i) {
    static_for<RJ>([&](auto rc) { constexpr int r = rc; ring[0][NP - 1][r] = un[r]; });
    if constexpr (NIN > 1)
      static_for<NIN - 1>([&](auto nc) { constexpr int n = nc; static_for<RJ>([&](auto rc) { constexpr int r = rc; fx[n][NQ - 1][r] = fn[n][r]; }); });
    // J-halo rows of every stage's centre plane: publish my first / last R1 own rows, take the neighbouring waves'
    const int buf = (i - i0) & 1;
    static_for<NS>([&](auto kc) {
      constexpr int k = kc;
      static_for<R1>([&](auto xc) {
        constexpr int x = xc;
        lds[buf][w][k][x][lane] = ring[k][R0][x];
        lds[buf][w][k][R1 + x][lane] = ring[k][R0][RJ - R1 + x];
      });
    });
    __syncthreads();
    V above[NS][R1], below[NS][R1];
    static_for<NS>([&](auto kc) {
      constexpr int k = kc;
      static_for<R1>([&](auto xc) {
        constexpr int x = xc;
        above[k][x] = ring[k][R0][0];        // window-edge waves: any value (those rows are not kept)
        below[k][x] = ring[k][R0][RJ - 1];
        if (w > 0) above[k][x] = lds[buf][w - 1][k][R1 + x][lane];      // row j0 - R1 + x = the wave above's last R1 rows
        if (w < WJ - 1) below[k][x] = lds[buf][w + 1][k][x][lane];      // row j0 + RJ + x = the wave below's first R1 rows
      });
    });
    if (i + NS * R0 + 1 <= ie - 1 + NS * R0) {   // u(ie - 1 + NS R0) is the last plane a kept result depends on
      load_plane(P.in[0], i + NS * R0 + 1, un);
      if constexpr (NIN > 1) static_for<NIN - 1>([&](auto nc) { constexpr int n = nc; load_plane(P.in[n + 1], i + NQ, fn[n]); });
    }
    static_for<NS>([&](auto kc) {
      constexpr int k = kc;                // stage k + 1: v_{k+1}(i + (NS - k - 1) R0) from ring[k]
      constexpr int q = (NS - k - 1) * R0;
      if constexpr (k + 1 < NS) stage(ring[k], above[k], below[k], fx, std::integral_constant<int, q>{}, i + q, ring[k + 1][NP - 1]);
      else stage(ring[k], above[k], below[k], fx, std::integral_constant<int, 0>{}, i, wres);
    });
    if (i >= ib && lane_keep) {
      char* obase = reinterpret_cast<char*>(P.out) + (int64_t)i * plane_b;
      static_for<RJ>([&](auto rc) {
        constexpr int r = rc;
        if (row_keep[r]) __builtin_nontemporal_store(wres[r], reinterpret_cast<V*>(obase + (rowb[r] + lane_b)));
      });
    }
    static_for<NS>([&](auto kc) {
      constexpr int k = kc;
      static_for<NP - 1>([&](auto pc) { constexpr int pp = pc; static_for<RJ>([&](auto rc) { constexpr int r = rc; ring[k][pp][r] = ring[k][pp + 1][r]; }); });
    });
    if constexpr (NIN > 1)
      static_for<NIN - 1>([&](auto nc) {
        constexpr int n = nc;
        static_for<NQ - 1>([&](auto qc) { constexpr int q = qc; static_for<RJ>([&](auto rc) { constexpr int r = rc; fx[n][q][r] = fx[n][q + 1][r]; }); });
      });
  }
}

// ---- rank 2 ------------------------------------------------------------------------------------------------
// The same idea for 2-D fields (d0, d1) -> (I, K): a wave marches DOWN THE ROWS of its column window (one wave span wide,
// the kept columns whole 64-byte granules as above) with three rows of every stage's input in registers.  There is no J
// axis, so waves are independent: no LDS, no barrier; PF rows stay in flight per wave instead.  Two or three time steps
// of a 2-D explicit scheme (the reference's own time-stepping inputs are 1-D / 2-D) cost one read and one write of the
// field plus 128/120 in columns and 2 NS rows per chunk.
struct March2R2Params {
  const void* in[NEPTUNE_HIP_MAX_INPUTS];   // in[0]: the field the applies chain on; in[1..]: centre-only inputs, the same at every stage
  void* out;
  int32_t N0, N1;          // rows, columns
  int32_t plb[2], pub[2];
  int64_t olb[2];
  int32_t rI0, rI1;        // rows this launch stores
  int32_t chunk;
  uint32_t nK;
};

// star footprints of input 0 up to radius 2 along the rows (R0) and the columns (R2); FP::R1 == 0 in rank 2
template <class FP>
constexpr bool march2_rank2_footprint() {
  return FP::MARCH_OK && !FP::BOX && FP::HALO_MASK == 1u && FP::R0 >= 1 && FP::R0 <= 2 && FP::R1 == 0 && FP::R2 >= 1 && FP::R2 <= 2;
}

template <class Body, class T, int NIN, class FP, int NS, int PF>
__global__ __launch_bounds__(256) void neptune_apply_march2_rank2(March2R2Params P, Body body) {
  using V = typename Vec16<T>::type;
  using GM = March2Geom<T, FP, NS>;
  constexpr int VK = GM::VK, R0 = GM::R0, R2 = GM::R2, NP = 2 * R0 + 1, MK = GM::MK, SPAN = GM::SPAN, KEEPK = GM::KEEPK;
  static_assert(NS >= 2 && NS <= 3 && march2_rank2_footprint<FP>() && R2 <= VK, "window constants");
  constexpr int NQ = (NS - 1) * R0 + 1;       // rows of a centre-only input between the first stage's row and the last one's
  constexpr int NF = NIN > 1 ? NIN - 1 : 1;
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t gw = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x >> 6);   // global wave id: column windows fastest
  const uint32_t kt = gw % P.nK, ct = gw / P.nK;
  const int32_t kw = (int32_t)(kt * KEEPK) - MK;
  const int32_t k0 = kw + lane * VK;
  const int32_t kc = k0 < 0 ? 0 : (k0 > P.N1 - VK ? P.N1 - VK : k0);
  const int32_t ib = P.rI0 + (int32_t)ct * P.chunk;
  const int32_t ie = (ib + P.chunk < P.rI1) ? ib + P.chunk : P.rI1;
  if (ib >= ie) return;
  T* out = static_cast<T*>(P.out);
  auto load_row = [&](int n, int32_t ip) -> V {
    const int32_t ic = ip < 0 ? 0 : (ip >= P.N0 ? P.N0 - 1 : ip);
    return *reinterpret_cast<const V*>(static_cast<const T*>(P.in[n]) + (int64_t)ic * P.N1 + kc);
  };
  bool in_k[VK];
  static_for<VK>([&](auto ec) { constexpr int e = ec; in_k[e] = (k0 + e) >= P.plb[1] && (k0 + e) < P.pub[1]; });
  const bool lane_keep = k0 >= kw + MK && k0 < kw + SPAN - MK && k0 >= 0 && k0 < P.N1;

  auto stage = [&](const V(&pl)[NP], const V(&fx)[NF][NQ], auto qc, int32_t ip) -> V {
    constexpr int q = decltype(qc)::value;
    V ring[1][NP][1];
    T lft[1][1][1][R2], rgt[1][1][1][R2];
    V pt[NIN][1];
    static_for<NP>([&](auto pc) { constexpr int pp = pc; ring[0][pp][0] = pl[pp]; });
    static_for<R2>([&](auto xc) {
      constexpr int x = xc;
      constexpr int dl = R2 - x, dr = x + 1;
      lft[0][0][0][x] = from_prev<true>(pl[R0][VK - dl], pl[R0][0], lane);
      rgt[0][0][0][x] = from_next<true>(pl[R0][dr - 1], pl[R0][VK - 1], lane);
    });
    static_for<NIN>([&](auto nc) { constexpr int n = nc; if constexpr (n > 0) pt[n][0] = fx[n - 1][q]; });
    const bool in_i = ip >= P.plb[0] && ip < P.pub[0];
    const int64_t li = (int64_t)ip + P.olb[0];
    V res;
    static_for<VK>([&](auto ec) {
      constexpr int e = ec;
      const int64_t lk = (int64_t)(k0 + e) + P.olb[1];
      MarchAcc<T, 2, NIN, FP, 1, 0, e, false> acc{ring, lft, rgt, pt, li, 0, lk};
      const T val = body(acc);
      res[e] = (in_i && in_k[e]) ? val : OutsideOf<Body, T>::apply(body, pl[R0][e]);
    });
    return res;
  };

  // ring[k]: the 2 R0 + 1 newest rows of stage input k (k = 0: the field itself); un: rows in flight; fx / fn: the centre-only
  // inputs' rows between the stages, and in flight
  V ring[NS][NP];
  V un[PF];
  V fx[NF][NQ], fn[NF][PF];
  const int32_t i0 = ib - GM::WARM;
  static_for<NP - 1>([&](auto pc) { constexpr int pp = pc; ring[0][pp] = load_row(0, i0 + NS * R0 - 2 * R0 + pp); });
  static_for<PF>([&](auto dc) { constexpr int d = dc; un[d] = load_row(0, i0 + NS * R0 + d); });
  if constexpr (NIN > 1) {
    static_for<NIN - 1>([&](auto nc) {
      constexpr int n = nc;
      static_for<NQ - 1>([&](auto qc) { constexpr int q = qc; fx[n][q] = load_row(n + 1, i0 + q); });
      static_for<PF>([&](auto dc) { constexpr int d = dc; fn[n][d] = load_row(n + 1, i0 + NQ - 1 + d); });
    });
  }
  static_for<NS - 1>([&](auto kc2) { constexpr int k = kc2 + 1; static_for<NP - 1>([&](auto pc) { constexpr int pp = pc; ring[k][pp] = ring[0][0]; }); });
  auto step = [&](int32_t i, auto slot_c) {
    constexpr int slot = slot_c;
    ring[0][NP - 1] = un[slot];
    if constexpr (NIN > 1) static_for<NIN - 1>([&](auto nc) { constexpr int n = nc; fx[n][NQ - 1] = fn[n][slot]; });
    if (i + NS * R0 + PF <= ie - 1 + NS * R0) {
      un[slot] = load_row(0, i + NS * R0 + PF);
      if constexpr (NIN > 1) static_for<NIN - 1>([&](auto nc) { constexpr int n = nc; fn[n][slot] = load_row(n + 1, i + NQ - 1 + PF); });
    }
    V w;
    static_for<NS>([&](auto kc2) {
      constexpr int k = kc2;
      constexpr int q = (NS - k - 1) * R0;
      if constexpr (k + 1 < NS) ring[k + 1][NP - 1] = stage(ring[k], fx, std::integral_constant<int, q>{}, i + q);
      else w = stage(ring[k], fx, std::integral_constant<int, 0>{}, i);
    });
    if (i >= ib && lane_keep) __builtin_nontemporal_store(w, reinterpret_cast<V*>(out + (int64_t)i * P.N1 + kc));
    static_for<NS>([&](auto kc2) { constexpr int k = kc2; static_for<NP - 1>([&](auto pc) { constexpr int pp = pc; ring[k][pp] = ring[k][pp + 1]; }); });
    if constexpr (NIN > 1)
      static_for<NIN - 1>([&](auto nc) { constexpr int n = nc; static_for<NQ - 1>([&](auto qc) { constexpr int q = qc; fx[n][q] = fx[n][q + 1]; }); });
  };
  for (int32_t i = i0; i < ie; i += PF) {
    static_for<PF>([&](auto phc) {
      constexpr int ph = phc;
      if (i + ph < ie) step(i + ph, phc);
    });
  }
}

template <class T, int NIN, class FP>
inline bool march2_rank2_eligible(const neptune_hip_apply_geom_t* g, const void* const* in, const void* out) {
  constexpr int G = 64 / (int)sizeof(T);
  if (!g || g->rank != 2 || g->num_inputs != NIN) return false;
  if (!march2_rank2_footprint<FP>()) return false;
  int64_t n[2];
  const int r[2] = {FP::R0, FP::R2};
  for (int d = 0; d < 2; ++d) {
    n[d] = g->out_ub[d] - g->out_lb[d];
    for (int k = 0; k < NIN; ++k)
      if (g->in_lb[k][d] != g->out_lb[d] || g->in_ub[k][d] != g->out_ub[d]) return false;
    if (g->lb[d] < g->ub[d] && (g->lb[d] - r[d] < g->out_lb[d] || g->ub[d] + r[d] > g->out_ub[d])) return false;
  }
  if (g->region_lb[1] != 0 || g->region_ub[1] != n[1]) return false;
  if (n[1] % G != 0 || n[1] < 2 * G || n[0] < 1 || n[0] >= 0x7fffffffLL || n[1] >= 0x7fffffffLL) return false;
  if ((uintptr_t)out % 64 != 0) return false;
  for (int k = 0; k < NIN; ++k)
    if (!in[k] || (uintptr_t)in[k] % 64 != 0) return false;
  return true;
}

template <class Body, class T, int NIN, class FP, int NS>
inline int launch_march2_rank2(const Body& body, const neptune_hip_apply_geom_t* g, const void* const* in, void* out, hipStream_t stream,
                               int chunk_req) {
  if (!march2_rank2_eligible<T, NIN, FP>(g, in, out) || geom_bounds_empty(g)) return NEPTUNE_HIP_EUNSUPPORTED;
  constexpr int KEEPK = March2Geom<T, FP, NS>::KEEPK, PF = 4;
  March2R2Params P{};
  for (int k = 0; k < NIN; ++k) P.in[k] = in[k];
  P.out = out;
  P.N0 = (int32_t)(g->out_ub[0] - g->out_lb[0]);
  P.N1 = (int32_t)(g->out_ub[1] - g->out_lb[1]);
  for (int d = 0; d < 2; ++d) {
    P.plb[d] = (int32_t)(g->lb[d] - g->out_lb[d]);
    P.pub[d] = (int32_t)(g->ub[d] - g->out_lb[d]);
    P.olb[d] = g->out_lb[d];
  }
  P.rI0 = (int32_t)g->region_lb[0];
  P.rI1 = (int32_t)g->region_ub[0];
  if (P.rI1 <= P.rI0) return NEPTUNE_HIP_OK;
  P.nK = (uint32_t)((P.N1 + KEEPK - 1) / KEEPK);
  const int64_t rows = P.rI1 - P.rI0;
  int64_t chunk = chunk_req > 0 ? chunk_req : 256;
  // enough waves for the chip (256 CUs x 32 waves), but never chunks so short that the warm-up rows dominate
  while (chunk_req <= 0 && chunk > 32 && (int64_t)P.nK * ((rows + chunk - 1) / chunk) < 8192) chunk /= 2;
  if (chunk > rows) chunk = rows;
  P.chunk = (int32_t)chunk;
  const int64_t waves = (int64_t)P.nK * ((rows + chunk - 1) / chunk);
  const int64_t blocks = (waves + 3) / 4;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return NEPTUNE_HIP_EUNSUPPORTED;
  hipLaunchKernelGGL((neptune_apply_march2_rank2<Body, T, NIN, FP, NS, PF>), dim3((uint32_t)blocks), dim3(256), 0, stream, P, body);
  NEPTUNE_HIP_CHECK(hipGetLastError());
  return NEPTUNE_HIP_OK;
}

// host side: can this geometry take the chain kernel, and launch it
template <class FP>
constexpr bool march2_footprint() {
  return FP::MARCH_OK && !FP::BOX && FP::HALO_MASK == 1u && FP::R0 >= 1 && FP::R0 <= 2 && FP::R1 >= 1 && FP::R1 <= 2 && FP::R2 >= 1 && FP::R2 <= 2;
}
template <class T, int NIN, class FP>
inline bool march2_eligible(const neptune_hip_apply_geom_t* g, const void* const* in, const void* out) {
  constexpr int G = 64 / (int)sizeof(T);
  if (!g || g->rank != 3 || g->num_inputs != NIN) return false;
  if (!march2_footprint<FP>()) return false;
  int64_t n[3];
  for (int d = 0; d < 3; ++d) {
    n[d] = g->out_ub[d] - g->out_lb[d];
    for (int k = 0; k < NIN; ++k)
      if (g->in_lb[k][d] != g->out_lb[d] || g->in_ub[k][d] != g->out_ub[d]) return false;
    if (d > 0 && (g->region_lb[d] != 0 || g->region_ub[d] != n[d])) return false;
  }
  if (n[2] % G != 0 || n[2] < 2 * G || n[1] < 8 || n[0] < 1) return false;   // rows = whole 64-byte granules
  if (n[1] * n[2] * (int64_t)sizeof(T) >= 0x7fffffffLL) return false;         // 32-bit in-plane offsets
  if ((uintptr_t)out % 64 != 0) return false;
  for (int k = 0; k < NIN; ++k)
    if (!in[k] || (uintptr_t)in[k] % 64 != 0) return false;
  // the second apply re-reads what the first one copied through: only sound when every access of an in-bounds cell
  // stays inside the box (the same rule every plan obeys)
  const int r[3] = {FP::R0, FP::R1, FP::R2};
  for (int d = 0; d < 3; ++d)
    if (g->lb[d] < g->ub[d] && (g->lb[d] - r[d] < g->out_lb[d] || g->ub[d] + r[d] > g->out_ub[d])) return false;
  return true;
}

template <class Body, class T, int NIN, class FP, int NS, int RJ, int WJ, int MINW>
inline int launch_march2_shape(const Body& body, const neptune_hip_apply_geom_t* g, const void* const* in, void* out, hipStream_t stream,
                               int chunk_req) {
  using GM = March2Geom<T, FP, NS>;
  constexpr int KEEPJ = RJ * WJ - 2 * GM::MJ, KEEPK = GM::KEEPK;
  March2Params<T, NIN> P{};
  for (int k = 0; k < NIN; ++k) P.in[k] = static_cast<const T*>(in[k]);
  P.out = static_cast<T*>(out);
  P.N0 = (int32_t)(g->out_ub[0] - g->out_lb[0]);
  P.N1 = (int32_t)(g->out_ub[1] - g->out_lb[1]);
  P.N2 = (int32_t)(g->out_ub[2] - g->out_lb[2]);
  for (int d = 0; d < 3; ++d) {
    P.plb[d] = (int32_t)(g->lb[d] - g->out_lb[d]);
    P.pub[d] = (int32_t)(g->ub[d] - g->out_lb[d]);
    P.olb[d] = g->out_lb[d];
  }
  P.rI0 = (int32_t)g->region_lb[0];
  P.rI1 = (int32_t)g->region_ub[0];
  if (P.rI1 <= P.rI0) return NEPTUNE_HIP_OK;
  P.nJ = (uint32_t)((P.N1 + KEEPJ - 1) / KEEPJ);
  P.nK = (uint32_t)((P.N2 + KEEPK - 1) / KEEPK);
  const int64_t planes = P.rI1 - P.rI0;
  int64_t chunk = chunk_req > 0 ? chunk_req : 128;
  // small fields: enough workgroups for every CU
  while (chunk_req <= 0 && chunk > 16 && (int64_t)P.nJ * P.nK * ((planes + chunk - 1) / chunk) < 512) chunk /= 2;
  if (chunk > planes) chunk = planes;
  P.chunk = (int32_t)chunk;
  const int64_t blocks = (int64_t)P.nJ * P.nK * ((planes + chunk - 1) / chunk);
  if (blocks <= 0 || blocks > 0x7fffffffLL) return NEPTUNE_HIP_EUNSUPPORTED;
  hipLaunchKernelGGL((neptune_apply_march2<Body, T, NIN, FP, NS, RJ, WJ, MINW>), dim3((uint32_t)blocks), dim3(kWave * WJ), 0, stream, P,
                     body);
  NEPTUNE_HIP_CHECK(hipGetLastError());
  return NEPTUNE_HIP_OK;
}

template <class Body, class T, int NIN, class FP, int NS>
inline int launch_march2(const Body& body, const neptune_hip_apply_geom_t* g, const void* const* in, void* out, hipStream_t stream,
                         int chunk_req) {
  if (!march2_eligible<T, NIN, FP>(g, in, out)) return NEPTUNE_HIP_EUNSUPPORTED;
  if (geom_bounds_empty(g)) return NEPTUNE_HIP_EUNSUPPORTED;   // pure copies: leave to the plain path
  // window shape: NEPTUNE_HIP_MARCH2 = 0..3 picks one for measurements (tools/twostep_bench.py)
  static const int shape = [] { const char* e = getenv("NEPTUNE_HIP_MARCH2"); return e ? atoi(e) : 0; }();
  constexpr bool wide = FP::R0 > 1 || FP::R1 > 1 || FP::R2 > 1;
  if constexpr (wide) {
    // radius 2 (13-point 4th-order operators): five planes per ring and two halo rows per side and stage; two applies per
    // pass only -- three rings of five planes do not fit the registers.  Eight waves (up to 256 VGPRs each):
    //   rows per lane x waves   4x8: window 32 rows, keeps 32 - 4 R1   3x8: 24 rows (fewer registers)
    if constexpr (NS != 2) {
      return NEPTUNE_HIP_EUNSUPPORTED;
    } else {
      // measured (512^3 fp64, profiles/r03_chain_wide.txt): 13-point Laplacian 3x8: 1.12x the steps per second of one apply per
      // launch, 4x8 (236 VGPRs): 0.99x; with a coefficient field 3x8: 1.18x, 4x8 spills: 0.45x
      switch (shape) {
        default:
        case 0: return launch_march2_shape<Body, T, NIN, FP, 2, 3, 8, 1>(body, g, in, out, stream, chunk_req);
        case 1: return launch_march2_shape<Body, T, NIN, FP, 2, 4, 8, 1>(body, g, in, out, stream, chunk_req);
        // (2x16 needs 136 VGPRs per wave: a 16-wave workgroup may use 128; 5x8 spills)
      }
    }
  } else if constexpr (NS == 2) {
    // Measured on 1024^3 fp64, 7-point operator, 128-plane chunks, steps/s against one apply per pass (2.82 ms/step),
    // profiles/r02_twostep.txt:  rows per lane x waves  3x16: 1.79x (108 VGPRs)   7x8: 1.81x (228 VGPRs)   6x8: 1.75x
    // 2x16: 1.74x   5x8: 1.75x   4x8: 1.59x   4x16 (spills): 1.51x.  Default: 3x16 -- sixteen waves keep 48 KiB of row
    // loads in flight per CU with registers to spare for bodies heavier than the Laplacian.
    switch (shape) {
      default:
      case 0: return launch_march2_shape<Body, T, NIN, FP, 2, 3, 16, 1>(body, g, in, out, stream, chunk_req);
#if NEPTUNE_HIP_FULL_VARIANTS
      case 1: return launch_march2_shape<Body, T, NIN, FP, 2, 7, 8, 1>(body, g, in, out, stream, chunk_req);
      case 2: return launch_march2_shape<Body, T, NIN, FP, 2, 4, 8, 1>(body, g, in, out, stream, chunk_req);
      case 3: return launch_march2_shape<Body, T, NIN, FP, 2, 2, 16, 1>(body, g, in, out, stream, chunk_req);
#endif
    }
  } else {
    // three applies per pass: three rings of three planes per lane
    switch (shape) {
      default:
      case 0: return launch_march2_shape<Body, T, NIN, FP, 3, 3, 12, 1>(body, g, in, out, stream, chunk_req);
#if NEPTUNE_HIP_FULL_VARIANTS
      case 1: return launch_march2_shape<Body, T, NIN, FP, 3, 2, 12, 1>(body, g, in, out, stream, chunk_req);
      case 2: return launch_march2_shape<Body, T, NIN, FP, 3, 4, 8, 1>(body, g, in, out, stream, chunk_req);
      case 3: return launch_march2_shape<Body, T, NIN, FP, 3, 5, 8, 1>(body, g, in, out, stream, chunk_req);
#endif
    }
  }
}

// what callers use: NS (2 or 3) chained applies of `body` in one pass if the footprint and the geometry allow it, else
// NEPTUNE_HIP_EUNSUPPORTED (the caller then launches the apply NS times); never instantiates the kernel for a footprint it
// cannot serve
template <class Body, class T, int RANK, int NIN, class FP, int NS = 2>
inline int launch_apply_chain(const Body& body, const neptune_hip_apply_geom_t* g, const void* const* in, void* out,
                              hipStream_t stream, const neptune_hip_launch_cfg_t* cfg) {
  if constexpr (RANK == 3 && march2_footprint<FP>()) {
    // rank 3: stars of input 0 up to radius 2 per axis; inputs 1.. (read at the centre only: FP::HALO_MASK == 1) are the same
    // field at every stage
    if (!g || !in || !out) return NEPTUNE_HIP_EINVAL;
    for (int k = 0; k < NIN; ++k)
      if (!in[k]) return NEPTUNE_HIP_EINVAL;
    if (cfg && cfg->kernel == NEPTUNE_HIP_KERNEL_DIRECT) return NEPTUNE_HIP_EUNSUPPORTED;
    const int rc = geom_validate(g);
    if (rc != NEPTUNE_HIP_OK) return rc;
    return launch_march2<Body, T, NIN, FP, NS>(body, g, in, out, stream, cfg ? cfg->chunk : 0);
  } else if constexpr (RANK == 2 && march2_rank2_footprint<FP>()) {
    // rank 2: stars of input 0 up to radius 2 along rows and columns (5- and 9-point operators), centre-only further inputs
    if (!g || !in || !out) return NEPTUNE_HIP_EINVAL;
    for (int k = 0; k < NIN; ++k)
      if (!in[k]) return NEPTUNE_HIP_EINVAL;
    if (cfg && cfg->kernel == NEPTUNE_HIP_KERNEL_DIRECT) return NEPTUNE_HIP_EUNSUPPORTED;
    const int rc = geom_validate(g);
    if (rc != NEPTUNE_HIP_OK) return rc;
    // (three rings of five rows: 60 VGPRs in fp64 -- fits, unlike rank 3)
    return launch_march2_rank2<Body, T, NIN, FP, NS>(body, g, in, out, stream, cfg ? cfg->chunk : 0);
  } else {
    return NEPTUNE_HIP_EUNSUPPORTED;
  }
}
template <class Body, class T, int RANK, int NIN, class FP>
inline int launch_apply_twice(const Body& body, const neptune_hip_apply_geom_t* g, const void* const* in, void* out,
                              hipStream_t stream, const neptune_hip_launch_cfg_t* cfg) {
  return launch_apply_chain<Body, T, RANK, NIN, FP, 2>(body, g, in, out, stream, cfg);
}
template <class Body, class T, int RANK, int NIN, class FP>
inline int launch_apply_thrice(const Body& body, const neptune_hip_apply_geom_t* g, const void* const* in, void* out,
                               hipStream_t stream, const neptune_hip_launch_cfg_t* cfg) {
  return launch_apply_chain<Body, T, RANK, NIN, FP, 3>(body, g, in, out, stream, cfg);
}

}  // namespace neptune_hip

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
// Scope: rank 3, one input, star footprint of radius 1 (the 7-point family, the fused explicit Euler step of such an
// operator included), all boxes equal, rows a whole number of 64-byte store granules.  Everything else keeps using
// two launches (neptune_hip_step_loop decides).
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

template <class T>
struct March2Params {
  const T* in;
  T* out;
  int32_t N0, N1, N2;
  int32_t plb[3], pub[3];  // apply.bounds, result-physical
  int64_t olb[3];          // logical origin (index arguments)
  int32_t rI0, rI1;        // planes this launch stores
  int32_t chunk;
  uint32_t nJ, nK;
};

// NS chained applies per pass (2 or 3); RJ rows per lane, WJ waves per workgroup, MINW = waves per SIMD the register
// allocation must leave room for.  Stage k = 1..NS computes v_k = A(v_{k-1}) (v_0 = the input u, v_NS = the result w);
// at step i stage k produces plane i + NS - k, so each stage's newest plane is the next stage's upper neighbour plane
// within the same step.  v_k is valid k cells / rows inside the window.
template <class Body, class T, class FP, int NS, int RJ, int WJ, int MINW>
__global__ __launch_bounds__(kWave* WJ, MINW) void neptune_apply_march2(March2Params<T> P, Body body) {
  using V = typename Vec16<T>::type;
  constexpr int VK = 16 / sizeof(T);
  static_assert(NS >= 2 && NS <= 3, "two or three applies per pass");
  constexpr int TJ = RJ * WJ;                 // window rows
  constexpr int KEEPJ = TJ - 2 * NS;          // rows of w this workgroup stores: [Jb + NS, Jb + TJ - NS)
  constexpr int G = 64 / (int)sizeof(T);      // cells per 64-byte store granule
  constexpr int SPAN = kWave * VK;            // window columns
  constexpr int KEEPK = SPAN - G;             // columns of w this wave stores: [kw + G/2, kw + SPAN - G/2)
  static_assert((G / 2) % VK == 0 && G / 2 >= NS, "the kept columns must start at a whole lane, NS cells inside");
  static_assert(FP::R0 == 1 && FP::R1 == 1 && FP::R2 == 1 && !FP::BOX && FP::HALO_MASK == 1u, "radius-1 star of input 0");
  __shared__ V lds[2][WJ][2 * NS][kWave];     // [parity][wave][stage input k: first own row, last own row][lane]

  const int lane = threadIdx.x & (kWave - 1);
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t vb = xcd_remap(blockIdx.x, gridDim.x);
  const uint32_t kt = vb % P.nK;
  const uint32_t t = vb / P.nK;
  const uint32_t jt = t % P.nJ;
  const uint32_t ct = t / P.nJ;

  const int32_t Jb = (int32_t)(jt * KEEPJ) - NS;         // first row of the window
  const int32_t j0 = Jb + w * RJ;                        // first own row
  const int32_t kw = (int32_t)(kt * KEEPK) - G / 2;      // first column of the window
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

  auto load_plane = [&](int32_t ip, V(&rows)[RJ]) {
    const int32_t ic = ip < 0 ? 0 : (ip >= P.N0 ? P.N0 - 1 : ip);
    const char* base = reinterpret_cast<const char*>(P.in) + (int64_t)ic * plane_b;
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
    row_keep[r] = j >= Jb + NS && j < Jb + TJ - NS && j >= 0 && j < P.N1;
    in_j[r] = j >= P.plb[1] && j < P.pub[1];
  });
  static_for<VK>([&](auto ec) {
    constexpr int e = ec;
    in_k[e] = (k0 + e) >= P.plb[2] && (k0 + e) < P.pub[2];
  });
  const bool lane_keep = k0 >= kw + G / 2 && k0 < kw + SPAN - G / 2 && k0 >= 0 && k0 < P.N2;

  // one stage: `ctr` = the centre plane's own rows, `above` / `below` its J-halo rows, lo / hi the planes before / after
  auto stage = [&](const V(&lo)[RJ], const V(&ctr)[RJ], const V(&hi)[RJ], const V& above, const V& below, int32_t ip,
                   V(&res)[RJ]) {
    V ring[1][3][RJ + 2];
    T lft[1][1][RJ + 2][1], rgt[1][1][RJ + 2][1];
    V pt[1][RJ];
    static_for<RJ>([&](auto rc) {
      constexpr int r = rc;
      ring[0][0][r + 1] = lo[r];
      ring[0][1][r + 1] = ctr[r];
      ring[0][2][r + 1] = hi[r];
      // K neighbours from the adjacent lanes; the wave's outermost cells get an arbitrary edge value: their results
      // are two cells outside the kept columns
      lft[0][0][r + 1][0] = from_prev<true>(ctr[r][VK - 1], ctr[r][0], lane);
      rgt[0][0][r + 1][0] = from_next<true>(ctr[r][0], ctr[r][VK - 1], lane);
    });
    ring[0][1][0] = above;
    ring[0][1][RJ + 1] = below;
    const bool in_i = ip >= P.plb[0] && ip < P.pub[0];
    const int64_t li = (int64_t)ip + P.olb[0];
    static_for<RJ>([&](auto rc) {
      constexpr int r = rc;
      const int64_t lj = (int64_t)(j0 + r) + P.olb[1];
      static_for<VK>([&](auto ec) {
        constexpr int e = ec;
        const int64_t lk = (int64_t)(k0 + e) + P.olb[2];
        MarchAcc<T, 3, 1, FP, RJ, r, e, false> acc{ring, lft, rgt, pt, li, lj, lk};
        const T val = body(acc);
        const T through = ctr[r][e];
        res[r][e] = (in_i && in_j[r] && in_k[e]) ? val : OutsideOf<Body, T>::apply(body, through);
      });
    });
  };

  // ---- register state.  ring[k] holds the three newest planes of v_k that stage k+1 reads: at step i these are planes
  // i+NS-k-2, i+NS-k-1 (the centre of stage k+1), i+NS-k; ring[k][2] is written by stage k in the same step (k = 0: by
  // the load issued one step earlier).
  V ring[NS][3][RJ];
  V un[RJ];                                // u(i + NS + 1) in flight
  V wres[RJ];
  const int32_t i0 = ib - 2 * (NS - 1);    // warm-up steps fill the rings of the intermediate fields (their w is discarded)
  load_plane(i0 + NS - 2, ring[0][0]);
  load_plane(i0 + NS - 1, ring[0][1]);
  load_plane(i0 + NS, un);
  static_for<NS - 1>([&](auto kc) {        // defined, never part of a kept result
    constexpr int k = kc + 1;
    static_for<RJ>([&](auto rc) { constexpr int r = rc; ring[k][0][r] = ring[0][0][r]; ring[k][1][r] = ring[0][0][r]; });
  });

  for (int32_t i = i0; i < ie; ++i) {
    static_for<RJ>([&](auto rc) { constexpr int r = rc; ring[0][2][r] = un[r]; });
    // J-halo rows of every stage's centre plane: publish my first / last own rows, take the neighbouring waves'
    const int buf = (i - i0) & 1;
    static_for<NS>([&](auto kc) {
      constexpr int k = kc;
      lds[buf][w][2 * k][lane] = ring[k][1][0];
      lds[buf][w][2 * k + 1][lane] = ring[k][1][RJ - 1];
    });
    __syncthreads();
    V above[NS], below[NS];
    static_for<NS>([&](auto kc) {
      constexpr int k = kc;
      above[k] = ring[k][1][0];            // window-edge waves: any value (those rows are not kept)
      below[k] = ring[k][1][RJ - 1];
      if (w > 0) above[k] = lds[buf][w - 1][2 * k + 1][lane];
      if (w < WJ - 1) below[k] = lds[buf][w + 1][2 * k][lane];
    });
    if (i + NS + 1 <= ie + NS - 1) load_plane(i + NS + 1, un);   // u(ie - 1 + NS) is the last plane a kept result depends on
    static_for<NS>([&](auto kc) {
      constexpr int k = kc;                // stage k + 1: v_{k+1}(i + NS - k - 1) from ring[k]
      if constexpr (k + 1 < NS) stage(ring[k][0], ring[k][1], ring[k][2], above[k], below[k], i + NS - k - 1, ring[k + 1][2]);
      else stage(ring[k][0], ring[k][1], ring[k][2], above[k], below[k], i, wres);
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
      static_for<RJ>([&](auto rc) { constexpr int r = rc; ring[k][0][r] = ring[k][1][r]; ring[k][1][r] = ring[k][2][r]; });
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
  const void* in;
  void* out;
  int32_t N0, N1;          // rows, columns
  int32_t plb[2], pub[2];
  int64_t olb[2];
  int32_t rI0, rI1;        // rows this launch stores
  int32_t chunk;
  uint32_t nK;
};

template <class Body, class T, class FP, int NS, int PF>
__global__ __launch_bounds__(256) void neptune_apply_march2_rank2(March2R2Params P, Body body) {
  using V = typename Vec16<T>::type;
  constexpr int VK = 16 / sizeof(T);
  constexpr int G = 64 / (int)sizeof(T), SPAN = kWave * VK, KEEPK = SPAN - G;
  static_assert(NS >= 2 && NS <= 3 && (G / 2) % VK == 0 && G / 2 >= NS, "window constants");
  static_assert(FP::R0 == 1 && FP::R1 == 0 && FP::R2 == 1 && !FP::BOX && FP::HALO_MASK == 1u, "radius-1 star of input 0, rank 2");
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t gw = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x >> 6);   // global wave id: column windows fastest
  const uint32_t kt = gw % P.nK, ct = gw / P.nK;
  const int32_t kw = (int32_t)(kt * KEEPK) - G / 2;
  const int32_t k0 = kw + lane * VK;
  const int32_t kc = k0 < 0 ? 0 : (k0 > P.N1 - VK ? P.N1 - VK : k0);
  const int32_t ib = P.rI0 + (int32_t)ct * P.chunk;
  const int32_t ie = (ib + P.chunk < P.rI1) ? ib + P.chunk : P.rI1;
  if (ib >= ie) return;
  const T* in = static_cast<const T*>(P.in);
  T* out = static_cast<T*>(P.out);
  auto load_row = [&](int32_t ip) -> V {
    const int32_t ic = ip < 0 ? 0 : (ip >= P.N0 ? P.N0 - 1 : ip);
    return *reinterpret_cast<const V*>(in + (int64_t)ic * P.N1 + kc);
  };
  bool in_k[VK];
  static_for<VK>([&](auto ec) { constexpr int e = ec; in_k[e] = (k0 + e) >= P.plb[1] && (k0 + e) < P.pub[1]; });
  const bool lane_keep = k0 >= kw + G / 2 && k0 < kw + SPAN - G / 2 && k0 >= 0 && k0 < P.N1;

  auto stage = [&](const V& lo, const V& ctr, const V& hi, int32_t ip) -> V {
    V ring[1][3][1] = {{{lo}, {ctr}, {hi}}};
    T lft[1][1][1][1], rgt[1][1][1][1];
    V pt[1][1];
    lft[0][0][0][0] = from_prev<true>(ctr[VK - 1], ctr[0], lane);
    rgt[0][0][0][0] = from_next<true>(ctr[0], ctr[VK - 1], lane);
    const bool in_i = ip >= P.plb[0] && ip < P.pub[0];
    const int64_t li = (int64_t)ip + P.olb[0];
    V res;
    static_for<VK>([&](auto ec) {
      constexpr int e = ec;
      const int64_t lk = (int64_t)(k0 + e) + P.olb[1];
      MarchAcc<T, 2, 1, FP, 1, 0, e, false> acc{ring, lft, rgt, pt, li, 0, lk};
      const T val = body(acc);
      res[e] = (in_i && in_k[e]) ? val : OutsideOf<Body, T>::apply(body, ctr[e]);
    });
    return res;
  };

  // ring[k]: the three newest rows of stage input k (k = 0: the field itself); un: rows in flight
  V ring[NS][3];
  V un[PF];
  const int32_t i0 = ib - 2 * (NS - 1);
  ring[0][0] = load_row(i0 + NS - 2);
  ring[0][1] = load_row(i0 + NS - 1);
  static_for<PF>([&](auto dc) { constexpr int d = dc; un[d] = load_row(i0 + NS + d); });
  static_for<NS - 1>([&](auto kc2) { constexpr int k = kc2 + 1; ring[k][0] = ring[0][0]; ring[k][1] = ring[0][0]; });
  auto step = [&](int32_t i, auto slot_c) {
    constexpr int slot = slot_c;
    ring[0][2] = un[slot];
    if (i + NS + PF <= ie + NS - 1) un[slot] = load_row(i + NS + PF);
    V w;
    static_for<NS>([&](auto kc2) {
      constexpr int k = kc2;
      if constexpr (k + 1 < NS) ring[k + 1][2] = stage(ring[k][0], ring[k][1], ring[k][2], i + NS - k - 1);
      else w = stage(ring[k][0], ring[k][1], ring[k][2], i);
    });
    if (i >= ib && lane_keep) __builtin_nontemporal_store(w, reinterpret_cast<V*>(out + (int64_t)i * P.N1 + kc));
    static_for<NS>([&](auto kc2) { constexpr int k = kc2; ring[k][0] = ring[k][1]; ring[k][1] = ring[k][2]; });
  };
  for (int32_t i = i0; i < ie; i += PF) {
    static_for<PF>([&](auto phc) {
      constexpr int ph = phc;
      if (i + ph < ie) step(i + ph, phc);
    });
  }
}

template <class T, class FP>
inline bool march2_rank2_eligible(const neptune_hip_apply_geom_t* g, const void* in, const void* out) {
  constexpr int G = 64 / (int)sizeof(T);
  if (!g || g->rank != 2 || g->num_inputs != 1) return false;
  if (!(FP::MARCH_OK && FP::R0 == 1 && FP::R1 == 0 && FP::R2 == 1 && !FP::BOX && FP::HALO_MASK == 1u)) return false;
  int64_t n[2];
  for (int d = 0; d < 2; ++d) {
    n[d] = g->out_ub[d] - g->out_lb[d];
    if (g->in_lb[0][d] != g->out_lb[d] || g->in_ub[0][d] != g->out_ub[d]) return false;
    if (g->lb[d] < g->ub[d] && (g->lb[d] - 1 < g->out_lb[d] || g->ub[d] + 1 > g->out_ub[d])) return false;
  }
  if (g->region_lb[1] != 0 || g->region_ub[1] != n[1]) return false;
  if (n[1] % G != 0 || n[1] < 2 * G || n[0] < 1 || n[0] >= 0x7fffffffLL || n[1] >= 0x7fffffffLL) return false;
  return (uintptr_t)in % 64 == 0 && (uintptr_t)out % 64 == 0;
}

template <class Body, class T, class FP, int NS>
inline int launch_march2_rank2(const Body& body, const neptune_hip_apply_geom_t* g, const void* in, void* out, hipStream_t stream,
                               int chunk_req) {
  if (!march2_rank2_eligible<T, FP>(g, in, out) || geom_bounds_empty(g)) return NEPTUNE_HIP_EUNSUPPORTED;
  constexpr int VK = 16 / (int)sizeof(T), G = 64 / (int)sizeof(T), KEEPK = kWave * VK - G, PF = 4;
  March2R2Params P{};
  P.in = in;
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
  // enough waves for the chip (256 CUs x 32 waves), but never chunks so short that the 2 NS warm-up rows dominate
  while (chunk_req <= 0 && chunk > 32 && (int64_t)P.nK * ((rows + chunk - 1) / chunk) < 8192) chunk /= 2;
  if (chunk > rows) chunk = rows;
  P.chunk = (int32_t)chunk;
  const int64_t waves = (int64_t)P.nK * ((rows + chunk - 1) / chunk);
  const int64_t blocks = (waves + 3) / 4;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return NEPTUNE_HIP_EUNSUPPORTED;
  hipLaunchKernelGGL((neptune_apply_march2_rank2<Body, T, FP, NS, PF>), dim3((uint32_t)blocks), dim3(256), 0, stream, P, body);
  NEPTUNE_HIP_CHECK(hipGetLastError());
  return NEPTUNE_HIP_OK;
}

// host side: can this geometry take the two-steps kernel, and launch it
template <class T, class FP>
inline bool march2_eligible(const neptune_hip_apply_geom_t* g, const void* in, const void* out) {
  constexpr int G = 64 / (int)sizeof(T);
  if (!g || g->rank != 3 || g->num_inputs != 1) return false;
  if (!(FP::MARCH_OK && FP::R0 == 1 && FP::R1 == 1 && FP::R2 == 1 && !FP::BOX && FP::HALO_MASK == 1u)) return false;
  int64_t n[3];
  for (int d = 0; d < 3; ++d) {
    n[d] = g->out_ub[d] - g->out_lb[d];
    if (g->in_lb[0][d] != g->out_lb[d] || g->in_ub[0][d] != g->out_ub[d]) return false;
    if (d > 0 && (g->region_lb[d] != 0 || g->region_ub[d] != n[d])) return false;
  }
  if (n[2] % G != 0 || n[2] < 2 * G || n[1] < 8 || n[0] < 1) return false;   // rows = whole 64-byte granules
  if (n[1] * n[2] * (int64_t)sizeof(T) >= 0x7fffffffLL) return false;         // 32-bit in-plane offsets
  if ((uintptr_t)in % 64 != 0 || (uintptr_t)out % 64 != 0) return false;
  // the second apply re-reads what the first one copied through: only sound when every access of an in-bounds cell
  // stays inside the box (the same rule every plan obeys)
  for (int d = 0; d < 3; ++d)
    if (g->lb[d] < g->ub[d] && (g->lb[d] - 1 < g->out_lb[d] || g->ub[d] + 1 > g->out_ub[d])) return false;
  return true;
}

template <class Body, class T, class FP, int NS, int RJ, int WJ, int MINW>
inline int launch_march2_shape(const Body& body, const neptune_hip_apply_geom_t* g, const void* in, void* out, hipStream_t stream,
                               int chunk_req) {
  constexpr int VK = 16 / (int)sizeof(T), G = 64 / (int)sizeof(T);
  constexpr int KEEPJ = RJ * WJ - 2 * NS, KEEPK = kWave * VK - G;
  March2Params<T> P{};
  P.in = static_cast<const T*>(in);
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
  hipLaunchKernelGGL((neptune_apply_march2<Body, T, FP, NS, RJ, WJ, MINW>), dim3((uint32_t)blocks), dim3(kWave * WJ), 0, stream, P,
                     body);
  NEPTUNE_HIP_CHECK(hipGetLastError());
  return NEPTUNE_HIP_OK;
}

template <class Body, class T, class FP, int NS>
inline int launch_march2(const Body& body, const neptune_hip_apply_geom_t* g, const void* in, void* out, hipStream_t stream,
                         int chunk_req) {
  if (!march2_eligible<T, FP>(g, in, out)) return NEPTUNE_HIP_EUNSUPPORTED;
  if (geom_bounds_empty(g)) return NEPTUNE_HIP_EUNSUPPORTED;   // pure copies: leave to the plain path
  // window shape: NEPTUNE_HIP_MARCH2 = 0..3 picks one for measurements (tools/twostep_bench.py)
  static const int shape = [] { const char* e = getenv("NEPTUNE_HIP_MARCH2"); return e ? atoi(e) : 0; }();
  if constexpr (NS == 2) {
    // Measured on 1024^3 fp64, 7-point operator, 128-plane chunks, steps/s against one apply per pass (2.82 ms/step),
    // profiles/r02_twostep.txt:  rows per lane x waves  3x16: 1.79x (108 VGPRs)   7x8: 1.81x (228 VGPRs)   6x8: 1.75x
    // 2x16: 1.74x   5x8: 1.75x   4x8: 1.59x   4x16 (spills): 1.51x.  Default: 3x16 -- sixteen waves keep 48 KiB of row
    // loads in flight per CU with registers to spare for bodies heavier than the Laplacian.
    switch (shape) {
      default:
      case 0: return launch_march2_shape<Body, T, FP, 2, 3, 16, 1>(body, g, in, out, stream, chunk_req);
#if NEPTUNE_HIP_FULL_VARIANTS
      case 1: return launch_march2_shape<Body, T, FP, 2, 7, 8, 1>(body, g, in, out, stream, chunk_req);
      case 2: return launch_march2_shape<Body, T, FP, 2, 4, 8, 1>(body, g, in, out, stream, chunk_req);
      case 3: return launch_march2_shape<Body, T, FP, 2, 2, 16, 1>(body, g, in, out, stream, chunk_req);
#endif
    }
  } else {
    // three applies per pass: three rings of three planes per lane
    switch (shape) {
      default:
      case 0: return launch_march2_shape<Body, T, FP, 3, 3, 12, 1>(body, g, in, out, stream, chunk_req);
#if NEPTUNE_HIP_FULL_VARIANTS
      case 1: return launch_march2_shape<Body, T, FP, 3, 2, 12, 1>(body, g, in, out, stream, chunk_req);
      case 2: return launch_march2_shape<Body, T, FP, 3, 4, 8, 1>(body, g, in, out, stream, chunk_req);
      case 3: return launch_march2_shape<Body, T, FP, 3, 5, 8, 1>(body, g, in, out, stream, chunk_req);
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
  if constexpr (RANK == 3 && NIN == 1 && FP::MARCH_OK && FP::R0 == 1 && FP::R1 == 1 && FP::R2 == 1 && !FP::BOX &&
                FP::HALO_MASK == 1u) {
    if (!g || !in || !in[0] || !out) return NEPTUNE_HIP_EINVAL;
    if (cfg && cfg->kernel == NEPTUNE_HIP_KERNEL_DIRECT) return NEPTUNE_HIP_EUNSUPPORTED;
    const int rc = geom_validate(g);
    if (rc != NEPTUNE_HIP_OK) return rc;
    return launch_march2<Body, T, FP, NS>(body, g, in[0], out, stream, cfg ? cfg->chunk : 0);
  } else if constexpr (RANK == 2 && NIN == 1 && FP::MARCH_OK && FP::R0 == 1 && FP::R1 == 0 && FP::R2 == 1 && !FP::BOX &&
                       FP::HALO_MASK == 1u) {
    if (!g || !in || !in[0] || !out) return NEPTUNE_HIP_EINVAL;
    if (cfg && cfg->kernel == NEPTUNE_HIP_KERNEL_DIRECT) return NEPTUNE_HIP_EUNSUPPORTED;
    const int rc = geom_validate(g);
    if (rc != NEPTUNE_HIP_OK) return rc;
    return launch_march2_rank2<Body, T, FP, NS>(body, g, in[0], out, stream, cfg ? cfg->chunk : 0);
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

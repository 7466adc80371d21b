// apply_plane.hpp -- rank-3 stencils whose neighbourhood does not fit the march kernel's registers: the march kernel's
// traversal with the J / K neighbourhood in LDS.  Two kernels:
//   neptune_apply_plane   STAR footprints (radius up to 8, one or several inputs read at offsets): the ring of the lane's own
//                         cells in registers, the centre plane's window in LDS
//   neptune_apply_planes  BOX footprints of radius 2: the windows of all live planes in LDS (second half of this file)
//
// A star reads its I neighbours (dim 0) on the cell's own (row, column) and its J / K neighbours on the centre plane only.
// The march kernel (apply_march.hpp) serves the J / K neighbours from registers: halo rows exchanged through LDS INTO
// registers, K neighbours by wave shifts with scalar halo cells.  For a 7-point stencil that is the cheapest form; for the
// high-order operators (radius 3-4: 19- and 25-point, the seismic stencils) the 2*R1 halo rows per lane, their in-flight
// copies and 2*R2 scalar halo cells per row leave room for TWO own rows per lane next to the 2*R0+1 ring planes -- 16 KiB
// of row loads in flight per CU, and the kernel is latency-bound at a third of the HBM rate (profiles/r01_highorder.txt).
//
// Here the registers hold nothing but, per input read at offsets, the ring of the lane's OWN cells (2*R0+1 planes x RJ rows
// x 16 B, never moved: addressed by a compile-time phase) and the planes in flight.  Every step the workgroup lays the plane
// that becomes the centre NEXT step out in LDS -- its own rows, the R1 rows above and below its window and the R2 cells left
// and right of it -- and the body reads J and K neighbours straight from there (ds_read with compile-time offsets; nothing
// staged in registers, no wave shifts, no scalar halo cells).  That frees the registers for FOUR own rows per lane at radius
// 4 and makes every radius up to 8 fit (RJ = 2).
//
//   * window    : WJ x WK waves, RJ rows x 64 lane vectors each; LDS holds [2][inputs][WJ*RJ + 2*R1][WK*64*VK + 2*HK] cells
//                 (double-buffered by step parity: ONE barrier per plane step; the writes of step i+1's window overlap the
//                 arithmetic of step i).
//   * halo rows : the 2*R1 rows outside the window are dealt over the workgroup's waves (one 16-byte row load per wave and
//                 step for radius 4 on the 8-wave tile) and requested two steps before their plane becomes the centre -- by
//                 then the neighbouring workgroup has pulled them through the XCD's L2 as its own rows.
//   * halo cells: the R2 cells beside the window's rows: one element load per row by the first 2*HK lanes of the
//                 window's outermost waves (clamped per cell, so ragged rows need nothing special).
//   * everything else (tile numbering, chunks of planes, clamped addresses, predicated non-temporal stores, copy-through
//     and bounds test folded into the store, ragged rows) is the march kernel's; it shares MarchParams and the launcher.
//
// Same semantics as the reference's loop nest (lib/Passes/DataflowLowering.cpp:258-448), same bits as the other kernels.
// Measurements: profiles/r02_plane.txt; design notes: DESIGN.md section 3.7.
#pragma once
#include "apply_march.hpp"

namespace neptune_hip {

// can the plane kernel run this footprint?  rank 3, star, radii that leave a window in LDS; several inputs read at offsets
// (systems of equations: h and q of a shallow-water residual at 4th order) each get a ring and a window of their own, as
// long as two rows per lane of all rings stay below ~170 VGPRs: NH * (2*max(R0,1)+2) <= 21 -- two inputs up to radius 4,
// three up to radius 2, four at radius 1
template <class FP>
constexpr int plane_slots() { return 2 * (FP::R0 > 1 ? FP::R0 : 1) + 2; }   // ring slots per halo input at PF = 1
template <class FP, int RANK>
constexpr bool plane_capable() {
  constexpr int NH = popcount_u(FP::HALO_MASK);
  return RANK == 3 && FP::MARCH_OK && !FP::BOX && NH >= 1 && NH * plane_slots<FP>() <= (NH == 1 ? 18 : 21) &&
         (FP::R1 > 0 || FP::R2 > 0) && FP::R0 <= 8 && FP::R1 <= 8 && FP::R2 <= 8;
}
// rows per lane the plane kernel gives a footprint on a WJ x WK window: the rings are NH * slots * RJ * 4 VGPRs (kept below
// ~170 of the 256 a wave has at two waves per SIMD), and the double-buffered windows must fit the CU's 160 KiB of LDS
template <class T, class FP>
constexpr int plane_rows(int rj, int wj, int wk) {
  constexpr int VK = 16 / (int)sizeof(T), HK = (FP::R2 + VK - 1) / VK * VK, NH = popcount_u(FP::HALO_MASK);
  while (rj > 1 && NH * plane_slots<FP>() * rj * 4 > 168) rj /= 2;
  while (rj > 1 && 2 * NH * (wj * rj + 2 * FP::R1) * (wk * kWave * VK + 2 * HK) * (int)sizeof(T) > 160 * 1024) rj /= 2;
  return rj;
}

// PH / NS / RR: the ring is addressed in place -- plane offset oi of the current step lives in slot (PH + oi + RR) mod NS
// (RR: planes the ring keeps on each side of the centre, max(R0, 1))
template <class T, int NIN, class FP, int RJ, int r, int e, int LROW, int WIN, int PH, int NS, int RR>
struct PlaneAcc {
  static constexpr int VK = 16 / sizeof(T);
  static constexpr int R0 = FP::R0, R1 = FP::R1, R2 = FP::R2, NP = 2 * R0 + 1;
  static constexpr unsigned HMASK = FP::HALO_MASK;
  using V = typename Vec16<T>::type;

  static constexpr int NH = popcount_u(HMASK);
  const V (&ring)[NH][NS][RJ];
  const V (&pt)[NIN][RJ];
  const T* lp;         // LDS: this lane's cell (own row 0, element 0) of the centre plane of halo input 0; input h: + h * WIN
  int64_t li, lj, lk;  // logical coordinates

  template <int IN, int... O>
  __device__ __forceinline__ T get() const {
    static_assert(IN >= 0 && IN < NIN, "input index out of range");
    constexpr int oi = PickOffset<3, 0, O...>::value, oj = PickOffset<3, 1, O...>::value, ok = PickOffset<3, 2, O...>::value;
    if constexpr ((HMASK >> IN) & 1u) {
      static_assert(oi >= -R0 && oi <= R0 && oj >= -R1 && oj <= R1 && ok >= -R2 && ok <= R2, "access outside the declared footprint");
      static_assert((oi != 0) + (oj != 0) + (ok != 0) <= 1, "star footprint declared but a diagonal access is used");
      // I neighbours and the J neighbours that are own rows of this lane sit in the ring; the rest is read from LDS
      constexpr int h = halo_slot(HMASK, IN);
      if constexpr (ok == 0 && r + oj >= 0 && r + oj < RJ) return ring[h][(PH + oi + RR) % NS][r + oj][e];
      else return lp[h * WIN + (r + oj) * LROW + e + ok];
    } else {
      static_assert(oi == 0 && oj == 0 && ok == 0, "only halo inputs may be read at an offset");
      return pt[IN][r][e];
    }
  }
  template <int D>
  __device__ __forceinline__ int64_t idx() const {
    static_assert(D >= 0 && D < 3, "index argument out of range");
    return D == 0 ? li : (D == 1 ? lj : lk);
  }
};

// TL: a march Tile (RJ, WJ, WK, PF, NT are used)
template <class Body, class T, int NIN, class FP, class TL>
__global__ __launch_bounds__(kWave* TL::WJ* TL::WK) void neptune_apply_plane(MarchParams<T, NIN> P, Body body) {
  constexpr int RJ = TL::RJ, WJ = TL::WJ, WK = TL::WK, PF = TL::PF, NW = WJ * WK;
  constexpr bool NT = TL::NT;
  using V = typename Vec16<T>::type;
  constexpr int VK = 16 / sizeof(T);
  constexpr int R0 = FP::R0, R1 = FP::R1, R2 = FP::R2;
  constexpr unsigned HMASK = FP::HALO_MASK;
  constexpr int NH = popcount_u(HMASK);       // inputs read at offsets: a ring and a window each
  static_assert(NH >= 1 && !FP::BOX, "plane kernel: star footprint");
  constexpr int HK = (R2 + VK - 1) / VK * VK;  // halo cells kept per side (whole lane vectors keep own cells 16-byte aligned)
  constexpr int TJ = WJ * RJ, SPAN = kWave * VK, TK = WK * SPAN;
  constexpr int LROW = TK + 2 * HK, LR = TJ + 2 * R1, WIN = LR * LROW;
  constexpr int NU = 2 * R1 * WK;              // halo-row units per input (one row x one wave span), all dealt over the waves
  constexpr int NUA = NH * NU;
  constexpr int NHW = NUA ? (NUA + NW - 1) / NW : 0, NHWX = NHW ? NHW : 1;
  static_assert(2 * HK <= kWave, "K halo cells are loaded by the first 2*HK lanes");
  static_assert(2 * NH * WIN * (int)sizeof(T) <= 160 * 1024, "windows do not fit the LDS");
  __shared__ __attribute__((aligned(16))) T lds[2][NH][LR][LROW];

  const int lane = threadIdx.x & (kWave - 1);
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wj = w / WK, wk = w % WK;

  const uint32_t v = xcd_remap(blockIdx.x, gridDim.x);
  const uint32_t kt = v % P.nK;
  const uint32_t t = v / P.nK;
  const uint32_t jt = t % P.nJ;
  const uint32_t ct = t / P.nJ;

  const int32_t j0w = P.rJ0 + (int32_t)(jt * TJ);   // the window's first row
  const int32_t j0 = j0w + wj * RJ;                 // first own row
  const int32_t kw0 = (int32_t)(kt * (uint32_t)TK);  // the window's first column
  const int32_t kw = kw0 + wk * SPAN;
  const int32_t k0 = kw + lane * VK;
  const bool lane_ok = k0 < P.Ks;
  const uint32_t lane_b = (uint32_t)(k0 < P.Kl ? k0 : P.Kl) * (uint32_t)sizeof(T);

  const int32_t ib = P.rI0 + (int32_t)ct * P.chunk;
  const int32_t ie = (ib + P.chunk < P.rI1) ? ib + P.chunk : P.rI1;
  if (ib >= ie) return;
  const int64_t plane_b = (int64_t)P.N1 * P.N2 * (int64_t)sizeof(T);

  // own-row offsets: one set for the result and input 0, one per input when there are several (InView, apply_march.hpp)
  constexpr int NV = NIN > 1 ? NIN : 1;
  constexpr auto vidx = [](int n) constexpr { return NIN > 1 ? n : 0; };
  uint32_t rowb[NV][RJ], lane_bv[NV];
  static_for<NV>([&](auto nc) {
    constexpr int n = nc;
    static_for<RJ>([&](auto rc) { constexpr int r = rc; rowb[n][r] = view_row_bytes(P, n, j0 + r); });
    lane_bv[n] = view_lane_bytes(P, n, k0);
  });

  // halo-row units of this wave: unit uu = w + t*NW -> halo input uu / NU, halo row x = (uu % NU) / WK (x < R1: above the
  // window), wave span (uu % NU) % WK.
  uint32_t hsrc[NHWX];      // byte offset within a plane of my 16 bytes of the unit
  int32_t hdst[NHWX];       // LDS cell index (within one buffer) they go to; -1: no unit
  int hin[NHWX];            // the unit's input
  bool hfix[NHWX];          // ... and whether my vector of it is the one InView::fix_k names
  static_for<NHW>([&](auto tc) {
    constexpr int tt = tc;
    const int uu = w + tt * NW;
    const int hh = NU ? uu / NU : 0, u = NU ? uu % NU : 0;
    const int x = u / WK, sp = u % WK;
    const int32_t kc = kw0 + sp * SPAN + lane * VK;
    hin[tt] = halo_input_of(HMASK, 0);
    static_for<NH>([&](auto hc) { constexpr int h = hc; if (hh == h) hin[tt] = halo_input_of(HMASK, h); });
    hsrc[tt] = view_row_bytes(P, hin[tt], x < R1 ? j0w - R1 + x : j0w + TJ + (x - R1)) + view_lane_bytes(P, hin[tt], kc);
    hdst[tt] = uu < NUA ? hh * WIN + (x < R1 ? x : TJ + x) * LROW + HK + sp * SPAN + lane * VK : -1;
    hfix[tt] = view_fix_star(P, hin[tt], kc);
  });
  // halo cells beside my own rows: lanes [0,HK) the cells left of the window (its leftmost waves), lanes [HK,2HK) the
  // cells right of it (its rightmost waves); clamped per cell
  const bool kh_left = R2 > 0 && wk == 0 && lane < HK, kh_right = R2 > 0 && wk == WK - 1 && lane >= HK && lane < 2 * HK;
  const bool kh_any = kh_left || kh_right;
  const int32_t khc = kh_left ? kw0 - HK + lane : kw0 + TK + (lane - HK);
  uint32_t kh_b[NV];
  static_for<NV>([&](auto nc) { constexpr int n = nc; kh_b[n] = view_cell_bytes(P, n, khc); });
  const int32_t kh_dst = (R1 + wj * RJ) * LROW + (kh_left ? lane : HK + TK + (lane - HK));

  // ---- register state: per halo input the ring of own cells with the planes in flight; the halo pieces in flight ----
  // A ring keeps RR = max(R0, 1) planes on each side of the centre (the plane AFTER the centre is laid out in LDS one
  // step ahead, so it must have arrived even when the body reads no I neighbour) plus the PF planes in flight: NS slots.
  // PHASED: the ring is never moved -- the step loop is unrolled NS times and step phase ph finds plane offset oi in slot
  // (ph + oi + RR) mod NS; the load of plane i+RR+PF goes straight into the slot of the plane that has just left the
  // stencil's reach.  (Rotating the ring costs (NP-1)*RJ 16-byte moves per step: 17 % of the vector instructions of a
  // radius-4 step.)  NS copies of the step must fit the instruction cache (64 KiB per two CUs; radius 4: 46 KiB; radius 5 with
  // three rows per lane was tried: 62 KiB, no faster than two rows and a rotated ring; radius 8 would be 110 KiB): rings beyond
  // 10 slots keep the rotation, slots NPR.. being the planes in flight then.
  constexpr int RR = R0 > 1 ? R0 : 1, NPR = 2 * RR + 1;
  constexpr int NS = NPR + PF;
  constexpr bool PHASED = NS <= 11;
  constexpr int UNROLL = PHASED ? NS : PF;
  V ring[NH][NS][RJ];
  V jh[NHWX];   // my halo-row units (and kh: the halo cells beside my rows) of plane i+1, in flight while step i-1 computes
  T kh[NH][RJ];
  V pt[NIN][RJ];   // inputs read at the centre only: row r of the next plane is requested as soon as row r has been computed

  auto load_rows = [&](int32_t ip, auto slot_c) {
    constexpr int sl = decltype(slot_c)::value;
    static_for<NH>([&](auto hc) {
      constexpr int h = hc, n = halo_input_of(HMASK, h);
      const char* base = view_plane_base(P, n, ip);
      static_for<RJ>([&](auto rc) { constexpr int r = rc; ring[h][sl][r] = *reinterpret_cast<const V*>(base + (rowb[vidx(n)][r] + lane_bv[vidx(n)])); });
    });
  };
  auto load_halos = [&](int32_t ip) {
    static_for<NHW>([&](auto tc) {
      constexpr int tt = tc;
      if (hdst[tt] >= 0) jh[tt] = *reinterpret_cast<const V*>(view_plane_base(P, hin[tt], ip) + hsrc[tt]);
    });
    if constexpr (R2 > 0) {
      if (kh_any) static_for<NH>([&](auto hc) {
        constexpr int h = hc, n = halo_input_of(HMASK, h);
        const char* base = view_plane_base(P, n, ip);
        static_for<RJ>([&](auto rc) { constexpr int r = rc; kh[h][r] = *reinterpret_cast<const T*>(base + (rowb[vidx(n)][r] + kh_b[vidx(n)])); });
      });
    }
  };
  auto load_point_row = [&](int32_t ip, auto rc) {
    constexpr int r = decltype(rc)::value;
    static_for<NIN>([&](auto nc) {
      constexpr int n = nc;
      if constexpr (!((HMASK >> n) & 1u)) pt[n][r] = *reinterpret_cast<const V*>(view_plane_base(P, n, ip) + (rowb[vidx(n)][r] + lane_bv[vidx(n)]));
    });
  };

  bool in_j[RJ], row_ok[RJ], in_k[VK];
  static_for<RJ>([&](auto rc) {
    constexpr int r = rc;
    in_j[r] = (j0 + r) >= P.plb[1] && (j0 + r) < P.pub[1];
    row_ok[r] = (j0 + r) < P.rJ1;
  });
  static_for<VK>([&](auto ec) {
    constexpr int e = ec;
    in_k[e] = (k0 + e) >= P.plb[2] && (k0 + e) < P.pub[2];
  });

  const int32_t own_cell = (R1 + wj * RJ) * LROW + HK + wk * SPAN + lane * VK;  // my cell (own row 0, element 0) in a window
  // lay a plane out in LDS buffer `b`: per halo input my own rows (from its ring slot), then my halo-row units and the halo
  // cells beside my rows
  auto lay_out = [&](int b, auto slot_c) {
    constexpr int sl = decltype(slot_c)::value;
    T* buf = &lds[b][0][0][0];
    // (an input in a box of its own with fewer than a lane vector of cells right of the result's rows: the lane there lays
    // its clamped vector out rotated into place, InView::fix_k / fix_d)
    static_for<NH>([&](auto hc) {
      constexpr int h = hc, n = halo_input_of(HMASK, h);
      static_for<RJ>([&](auto rc) {
        constexpr int r = rc;
        *reinterpret_cast<V*>(buf + h * WIN + own_cell + r * LROW) = view_fix<T>(ring[h][sl][r], view_fix_star(P, n, k0), view_fix_d(P, n));
      });
    });
    static_for<NHW>([&](auto tc) {
      constexpr int tt = tc;
      if (hdst[tt] >= 0) *reinterpret_cast<V*>(buf + hdst[tt]) = view_fix<T>(jh[tt], hfix[tt], view_fix_d(P, hin[tt]));
    });
    if constexpr (R2 > 0) {
      if (kh_any) static_for<NH>([&](auto hc) {
        constexpr int h = hc;
        static_for<RJ>([&](auto rc) { constexpr int r = rc; buf[h * WIN + kh_dst + r * LROW] = kh[h][r]; });
      });
    }
  };

  // ---- prologue: planes ib-RR .. ib+RR+PF-1 (the last PF of them stay in flight); plane ib laid out in buffer 0, the
  // halos of plane ib+1 in flight
  load_halos(ib);
  static_for<NS - 1>([&](auto pc) {
    constexpr int p = pc;
    // phased: plane ib-RR+p -> slot p.  rotating: slots 1.. (shifted down at the top of the first step)
    load_rows(ib - RR + p, std::integral_constant<int, PHASED ? p : p + 1>{});
  });
  static_for<RJ>([&](auto rc) { load_point_row(ib, rc); });
  lay_out(0, std::integral_constant<int, PHASED ? RR : RR + 1>{});
  load_halos(ib + 1);

  // One plane step.  Order: barrier (buffer i&1 is complete, the other one free) -> lay plane i+1 out in the other buffer
  // (its halos were requested a step ago) -> request the halos of plane i+2 and the rows of plane i+RR+PF -> compute
  // plane i from buffer i&1.  The LDS writes and every load thus have a whole compute phase to complete in.
  auto step = [&](const int32_t i, auto phase_c) {
    constexpr int ph = PHASED ? decltype(phase_c)::value : 0;   // ring phase
    constexpr int slot = decltype(phase_c)::value % PF;
    constexpr int CS = (ph + RR) % NS;                            // slot of the centre plane
    if constexpr (!PHASED) {
      static_for<NH>([&](auto hc) {
        constexpr int h = hc;
        static_for<NPR - 1>([&](auto pc) {
          constexpr int p = pc;
          static_for<RJ>([&](auto rc) { constexpr int r = rc; ring[h][p][r] = ring[h][p + 1][r]; });
        });
        static_for<RJ>([&](auto rc) { constexpr int r = rc; ring[h][NPR - 1][r] = ring[h][NPR + slot][r]; });
      });
    }

    __syncthreads();
    // no "is that plane still in my chunk?" tests here: plane indices are clamped into the field, so the few loads past
    // the chunk's end are harmless, and straight-line code lets the compiler count its waits exactly (with the tests it
    // drained vmcnt to zero at the top of every step)
    lay_out((i + 1 - ib) & 1, std::integral_constant<int, (CS + 1) % NS>{});
    load_halos(i + 2);
    load_rows(i + PF + RR, std::integral_constant<int, PHASED ? (ph + NS - 1) % NS : NPR + slot>{});

    const bool in_i = i >= P.plb[0] && i < P.pub[0];
    const int64_t li = (int64_t)i + P.olb[0];
    char* obase = reinterpret_cast<char*>(P.out) + (int64_t)i * plane_b;
    const T* lp = &lds[(i - ib) & 1][0][0][0] + own_cell;
    static_for<RJ>([&](auto rc) {
      constexpr int r = rc;
      const int64_t lj = (int64_t)(j0 + r) + P.olb[1];
      const bool in_ij = in_i && in_j[r];
      V res;
      static_for<VK>([&](auto ec) {
        constexpr int e = ec;
        const int64_t lk = (int64_t)(k0 + e) + P.olb[2];
        const bool inside = in_ij && in_k[e];
        PlaneAcc<T, NIN, FP, RJ, r, e, LROW, WIN, ph, NS, RR> acc{ring, pt, lp, li, lj, lk};
        const T val = body(acc);
        T through;
        if constexpr (HMASK & 1u) through = ring[0][CS][r][e];   // input 0 owns ring 0
        else through = pt[0][r][e];
        res[e] = inside ? val : OutsideOf<Body, T>::apply(body, through);
      });
      if (row_ok[r] && lane_ok) {
        V* dst = reinterpret_cast<V*>(obase + (rowb[0][r] + lane_b));
        if constexpr (NT) __builtin_nontemporal_store(res, dst);
        else *dst = res;
      }
      if constexpr (NIN > NH) load_point_row(i + 1, rc);
      // one row at a time: without the fence the scheduler hoists every row's LDS reads to the top of the step and the
      // 2*(R1+R2) neighbour vectors of ALL rows are live at once (radius 4: 120 VGPRs, spills)
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  for (int32_t i = ib; i < ie; i += UNROLL) {
    static_for<UNROLL>([&](auto phc) {
      constexpr int ph = phc;
      if (i + ph < ie) step(i + ph, phc);
    });
  }
}

// ---- BOX footprints: every live plane in LDS ------------------------------------------------------------------------
// A box stencil (27-point, 125-point) reads J / K neighbours on EVERY plane of its reach, so the window of every live
// plane lies in LDS -- NB = 2*R0 + 2 buffers: the 2*R0+1 planes the current step reads and the one being laid out for the
// next step -- and the registers hold nothing across steps but the loads in flight.  The march kernel serves a box from
// registers: K neighbours by a wave shift per plane, row and side, with scalar halo cells that have to ride the ring (the
// 27-point kernel spends 40 % of its vector instructions on moves, shifts and scalar-register spills, two waves per SIMD,
// profiles/r02_valu_27pt.txt).  Here a tap is an LDS read with a compile-time offset; the plane-to-buffer mapping is
// compile-time too (the step loop is unrolled NB times).  Traversal, halo-row units and halo cells as in the star kernel
// above; the halo-row units carry the window's corner cells as well.
template <class T, class FP>
constexpr int planes_rows(int rj, int wj, int wk) {
  constexpr int VK = 16 / (int)sizeof(T), HK = (FP::R2 + VK - 1) / VK * VK;
  while (rj > 1 && (2 * FP::R0 + 2) * (wj * rj + 2 * FP::R1) * (wk * kWave * VK + 2 * HK) * (int)sizeof(T) > 160 * 1024) rj /= 2;
  return rj;
}
template <class T, class FP, int RANK>
constexpr bool planes_capable() {
  constexpr int VK = 16 / (int)sizeof(T), HK = (FP::R2 + VK - 1) / VK * VK;
  return RANK == 3 && FP::MARCH_OK && FP::BOX && popcount_u(FP::HALO_MASK) == 1 && FP::R0 >= 1 && FP::R0 <= 2 && FP::R1 <= 2 &&
         FP::R2 <= 2 && (2 * FP::R0 + 2) * (4 + 2 * FP::R1) * (kWave * VK + 2 * HK) * (int)sizeof(T) <= 160 * 1024;
}

template <class T, int NIN, class FP, int RJ, int r, int e, int LROW, int PLANE, int PH, int NB>
struct PlanesAcc {
  static constexpr int R0 = FP::R0, R1 = FP::R1, R2 = FP::R2;
  static constexpr unsigned HMASK = FP::HALO_MASK;
  using V = typename Vec16<T>::type;
  const V (&pt)[NIN][RJ];
  const T* lp;         // LDS: this lane's cell (own row 0, element 0) in buffer 0
  int64_t li, lj, lk;

  template <int IN, int... O>
  __device__ __forceinline__ T get() const {
    static_assert(IN >= 0 && IN < NIN, "input index out of range");
    constexpr int oi = PickOffset<3, 0, O...>::value, oj = PickOffset<3, 1, O...>::value, ok = PickOffset<3, 2, O...>::value;
    if constexpr ((HMASK >> IN) & 1u) {
      static_assert(oi >= -R0 && oi <= R0 && oj >= -R1 && oj <= R1 && ok >= -R2 && ok <= R2, "access outside the declared footprint");
      // whole 16-byte vectors only: the cell k0+e+ok lies in my own vector or in a neighbouring lane's, and every tap of a
      // (plane, row) is served by the same two or three conflict-free ds_read_b128 (element reads at a 16-byte lane stride
      // are 4-way bank conflicts: ds_read(2)_b32 bank = (address / 4) mod 32)
      constexpr int VKc = 16 / (int)sizeof(T), q = e + ok;
      constexpr int vo = q >= 0 ? q / VKc : -((-q + VKc - 1) / VKc);
      const V vec = *reinterpret_cast<const V*>(lp + ((PH + oi + R0) % NB) * PLANE + (r + oj) * LROW + vo * VKc);
      return vec[q - vo * VKc];
    } else {
      static_assert(oi == 0 && oj == 0 && ok == 0, "only halo inputs may be read at an offset");
      return pt[IN][r][e];
    }
  }
  template <int D>
  __device__ __forceinline__ int64_t idx() const {
    static_assert(D >= 0 && D < 3, "index argument out of range");
    return D == 0 ? li : (D == 1 ? lj : lk);
  }
};

template <class Body, class T, int NIN, class FP, class TL>
__global__ __launch_bounds__(kWave* TL::WJ* TL::WK) void neptune_apply_planes(MarchParams<T, NIN> P, Body body) {
  constexpr int RJ = TL::RJ, WJ = TL::WJ, WK = TL::WK, PF = TL::PF, NW = WJ * WK;
  constexpr bool NT = TL::NT;
  using V = typename Vec16<T>::type;
  constexpr int VK = 16 / sizeof(T);
  constexpr int R0 = FP::R0, R1 = FP::R1, R2 = FP::R2, NP = 2 * R0 + 1, NB = NP + 1;
  constexpr unsigned HMASK = FP::HALO_MASK;
  static_assert(popcount_u(HMASK) == 1, "planes kernel: one halo input");
  constexpr int HIN = halo_input_of(HMASK, 0);
  constexpr int HK = (R2 + VK - 1) / VK * VK;
  constexpr int TJ = WJ * RJ, SPAN = kWave * VK, TK = WK * SPAN;
  constexpr int LROW = TK + 2 * HK, LR = TJ + 2 * R1, PLANE = LR * LROW;
  constexpr int NU = 2 * R1 * WK;
  constexpr int NHW = NU ? (NU + NW - 1) / NW : 0, NHWX = NHW ? NHW : 1;
  static_assert(2 * HK <= kWave, "K halo cells are loaded by the first 2*HK lanes");
  static_assert(NB * PLANE * (int)sizeof(T) <= 160 * 1024, "windows do not fit the LDS");
  __shared__ __attribute__((aligned(16))) T lds[NB][LR][LROW];

  const int lane = threadIdx.x & (kWave - 1);
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wj = w / WK, wk = w % WK;
  const uint32_t v = xcd_remap(blockIdx.x, gridDim.x);
  const uint32_t kt = v % P.nK;
  const uint32_t t = v / P.nK;
  const uint32_t jt = t % P.nJ;
  const uint32_t ct = t / P.nJ;
  const int32_t j0w = P.rJ0 + (int32_t)(jt * TJ);
  const int32_t j0 = j0w + wj * RJ;
  const int32_t kw0 = (int32_t)(kt * (uint32_t)TK);
  const int32_t kw = kw0 + wk * SPAN;
  const int32_t k0 = kw + lane * VK;
  const bool lane_ok = k0 < P.Ks;
  const uint32_t lane_b = (uint32_t)(k0 < P.Kl ? k0 : P.Kl) * (uint32_t)sizeof(T);
  const int32_t ib = P.rI0 + (int32_t)ct * P.chunk;
  const int32_t ie = (ib + P.chunk < P.rI1) ? ib + P.chunk : P.rI1;
  if (ib >= ie) return;
  const int64_t plane_b = (int64_t)P.N1 * P.N2 * (int64_t)sizeof(T);

  // own-row offsets: one set for the result and input 0, one per input when there are several (InView, apply_march.hpp)
  constexpr int NV = NIN > 1 ? NIN : 1;
  constexpr auto vidx = [](int n) constexpr { return NIN > 1 ? n : 0; };
  auto row_bytes = [&](int32_t j) -> uint32_t { return view_row_bytes(P, HIN, j); };   // rows of the input read at offsets
  uint32_t rowb[NV][RJ], lane_bv[NV];
  static_for<NV>([&](auto nc) {
    constexpr int n = nc;
    static_for<RJ>([&](auto rc) { constexpr int r = rc; rowb[n][r] = view_row_bytes(P, n, j0 + r); });
    lane_bv[n] = view_lane_bytes(P, n, k0);
  });
  // halo cells: lanes [0,HK) left of the window, lanes [HK,2HK) right of it (one cell each, clamped)
  const bool c_left = lane < HK, c_right = lane >= HK && lane < 2 * HK;
  const uint32_t cell_b = view_cell_bytes(P, HIN, c_left ? kw0 - HK + lane : kw0 + TK + (lane - HK));
  const int32_t cell_col = c_left ? lane : HK + TK + (lane - HK);
  const bool kh_any = R2 > 0 && ((wk == 0 && c_left) || (wk == WK - 1 && c_right));
  const int32_t kh_dst = (R1 + wj * RJ) * LROW + cell_col;
  // halo-row units of this wave (row x of the 2*R1 halo rows, wave span s), each with the window's corner cells beside it
  uint32_t hsrc[NHWX], hcsrc[NHWX];
  int32_t hdst[NHWX], hcdst[NHWX];
  bool hc_any[NHWX], hfix[NHWX];
  static_for<NHW>([&](auto tc) {
    constexpr int tt = tc;
    const int u = w + tt * NW;
    const int x = u / WK, sp = u % WK;
    const int32_t kc = kw0 + sp * SPAN + lane * VK;
    const uint32_t rb = row_bytes(x < R1 ? j0w - R1 + x : j0w + TJ + (x - R1));
    const int32_t lrow = (x < R1 ? x : TJ + x) * LROW;
    hsrc[tt] = rb + view_lane_bytes(P, HIN, kc);
    hfix[tt] = view_fix_star(P, HIN, kc);
    hdst[tt] = u < NU ? lrow + HK + sp * SPAN + lane * VK : -1;
    hcsrc[tt] = rb + cell_b;
    hcdst[tt] = lrow + cell_col;
    hc_any[tt] = R2 > 0 && u < NU && ((sp == 0 && c_left) || (sp == WK - 1 && c_right));
  });

  // loads in flight: slot d holds the pieces of one plane
  V nxt[PF][RJ];
  V jh[PF][NHWX];
  T kh[PF][RJ], hc[PF][NHWX];
  V pt[NIN][RJ];

  auto load_plane = [&](int32_t ip, auto dc) {
    constexpr int d = decltype(dc)::value;
    const char* base = view_plane_base(P, HIN, ip);
    static_for<NHW>([&](auto tc) {
      constexpr int tt = tc;
      if (hdst[tt] >= 0) jh[d][tt] = *reinterpret_cast<const V*>(base + hsrc[tt]);
      if (hc_any[tt]) hc[d][tt] = *reinterpret_cast<const T*>(base + hcsrc[tt]);
    });
    if constexpr (R2 > 0) {
      if (kh_any) static_for<RJ>([&](auto rc) { constexpr int r = rc; kh[d][r] = *reinterpret_cast<const T*>(base + (rowb[vidx(HIN)][r] + cell_b)); });
    }
    static_for<RJ>([&](auto rc) { constexpr int r = rc; nxt[d][r] = *reinterpret_cast<const V*>(base + (rowb[vidx(HIN)][r] + lane_bv[vidx(HIN)])); });
  };
  const int32_t own_cell = (R1 + wj * RJ) * LROW + HK + wk * SPAN + lane * VK;
  auto lay_out = [&](int b, auto dc) {
    constexpr int d = decltype(dc)::value;
    T* buf = &lds[0][0][0] + b * PLANE;
    static_for<RJ>([&](auto rc) {
      constexpr int r = rc;
      *reinterpret_cast<V*>(buf + own_cell + r * LROW) = view_fix<T>(nxt[d][r], view_fix_star(P, HIN, k0), view_fix_d(P, HIN));
    });
    static_for<NHW>([&](auto tc) {
      constexpr int tt = tc;
      if (hdst[tt] >= 0) *reinterpret_cast<V*>(buf + hdst[tt]) = view_fix<T>(jh[d][tt], hfix[tt], view_fix_d(P, HIN));
      if (hc_any[tt]) buf[hcdst[tt]] = hc[d][tt];
    });
    if constexpr (R2 > 0) {
      if (kh_any) static_for<RJ>([&](auto rc) { constexpr int r = rc; buf[kh_dst + r * LROW] = kh[d][r]; });
    }
  };
  auto load_point_row = [&](int32_t ip, auto rc) {
    constexpr int r = decltype(rc)::value;
    static_for<NIN>([&](auto nc) {
      constexpr int n = nc;
      if constexpr (n != HIN) pt[n][r] = *reinterpret_cast<const V*>(view_plane_base(P, n, ip) + (rowb[vidx(n)][r] + lane_bv[vidx(n)]));
    });
  };

  bool in_j[RJ], row_ok[RJ], in_k[VK];
  static_for<RJ>([&](auto rc) {
    constexpr int r = rc;
    in_j[r] = (j0 + r) >= P.plb[1] && (j0 + r) < P.pub[1];
    row_ok[r] = (j0 + r) < P.rJ1;
  });
  static_for<VK>([&](auto ec) {
    constexpr int e = ec;
    in_k[e] = (k0 + e) >= P.plb[2] && (k0 + e) < P.pub[2];
  });

  // ---- prologue: planes ib-R0 .. ib+R0 into buffers 0 .. 2*R0 (plane q lives in buffer (q - ib + R0) mod NB), then the
  // next PF planes in flight
  static_for<NP>([&](auto pc) {
    constexpr int p = pc;
    load_plane(ib - R0 + p, std::integral_constant<int, 0>{});
    lay_out(p, std::integral_constant<int, 0>{});
  });
  static_for<PF>([&](auto dc) { constexpr int d = dc; load_plane(ib + R0 + 1 + d, dc); });
  static_for<RJ>([&](auto rc) { load_point_row(ib, rc); });

  // one plane step at phase ph = (i - ib) mod NB: barrier (every buffer the step reads is complete, the buffer of plane
  // i-R0-1 free) -> lay plane i+R0+1 out there -> request plane i+R0+1+PF -> compute plane i
  auto step = [&](const int32_t i, auto phase_c) {
    constexpr int ph = decltype(phase_c)::value % NB;
    constexpr int slot = decltype(phase_c)::value % PF;
    __syncthreads();
    lay_out((ph + NB - 1) % NB, std::integral_constant<int, slot>{});
    load_plane(i + R0 + 1 + PF, std::integral_constant<int, slot>{});

    const bool in_i = i >= P.plb[0] && i < P.pub[0];
    const int64_t li = (int64_t)i + P.olb[0];
    char* obase = reinterpret_cast<char*>(P.out) + (int64_t)i * plane_b;
    const T* lp = &lds[0][0][0] + own_cell;
    static_for<RJ>([&](auto rc) {
      constexpr int r = rc;
      const int64_t lj = (int64_t)(j0 + r) + P.olb[1];
      const bool in_ij = in_i && in_j[r];
      V res;
      static_for<VK>([&](auto ec) {
        constexpr int e = ec;
        const int64_t lk = (int64_t)(k0 + e) + P.olb[2];
        const bool inside = in_ij && in_k[e];
        PlanesAcc<T, NIN, FP, RJ, r, e, LROW, PLANE, ph, NB> acc{pt, lp, li, lj, lk};
        const T val = body(acc);
        T through;
        if constexpr (HMASK & 1u) through = (*reinterpret_cast<const V*>(lp + ((ph + R0) % NB) * PLANE + r * LROW))[e];
        else through = pt[0][r][e];
        res[e] = inside ? val : OutsideOf<Body, T>::apply(body, through);
      });
      if (row_ok[r] && lane_ok) {
        V* dst = reinterpret_cast<V*>(obase + (rowb[0][r] + lane_b));
        if constexpr (NT) __builtin_nontemporal_store(res, dst);
        else *dst = res;
      }
      if constexpr (NIN > 1) load_point_row(i + 1, rc);
      // radius 1 (27 taps): no fence between rows -- adjacent own rows read the same LDS rows, and the compiler may keep them;
      // radius 2: one row at a time, or the 125 taps' vectors of all rows are live at once
      if constexpr (R0 > 1 || R1 > 1 || R2 > 1) __builtin_amdgcn_sched_barrier(0);
    });
  };

  constexpr int UNROLL = (NB % PF == 0) ? NB : NB * PF;   // phases and slots both stay compile-time
  for (int32_t i = ib; i < ie; i += UNROLL) {
    static_for<UNROLL>([&](auto phc) {
      constexpr int ph = phc;
      if (i + ph < ie) step(i + ph, phc);
    });
  }
}

// ---- RANK 2: the window of a tile in LDS ---------------------------------------------------------------------------
// 2-D footprints beyond what the march kernel's registers hold -- stars beyond radius 4, boxes beyond radius 2 (7x7, 9x9
// windows), several wide halo inputs -- in the shape of the rank-2 tile form (apply_march.hpp, Tile::JK2): the field is ONE
// plane tiled in (rows, columns), a workgroup loads its window once, computes and retires.  Here the whole window of every
// input read at offsets -- own rows, the R rows above and below, the halo cells and the corners -- goes to LDS and every
// tap is served from whole 16-byte LDS vectors (as in neptune_apply_planes); registers hold nothing but the loads in flight.
// (d0, d1) -> (J, K): FP::R0 is the row radius, FP::R2 the column radius.
template <class T, class FP, int RANK>
constexpr bool tile2_capable() {
  return RANK == 2 && FP::MARCH_OK && popcount_u(FP::HALO_MASK) >= 1 && FP::R0 <= 8 && FP::R2 <= 8;
}
template <class T, class FP>
constexpr int tile2_rows(int rj, int wj, int wk) {
  constexpr int VK = 16 / (int)sizeof(T), HK = (FP::R2 + VK - 1) / VK * VK, NH = popcount_u(FP::HALO_MASK);
  while (rj > 1 && NH * (wj * rj + 2 * FP::R0) * (wk * kWave * VK + 2 * HK) * (int)sizeof(T) > 64 * 1024) rj /= 2;   // two workgroups per CU
  return rj;
}

template <class T, int NIN, class FP, int RJ, int r, int e, int LROW, int WIN>
struct Tile2Acc {
  static constexpr int R1 = FP::R0, R2 = FP::R2, VK = 16 / (int)sizeof(T);
  static constexpr unsigned HMASK = FP::HALO_MASK;
  using V = typename Vec16<T>::type;
  const V (&pt)[NIN][RJ];
  const T* lp;         // LDS: this lane's cell (own row 0, element 0) in the window of halo input 0
  int64_t lj, lk;

  template <int IN, int... O>
  __device__ __forceinline__ T get() const {
    static_assert(IN >= 0 && IN < NIN, "input index out of range");
    constexpr int oj = PickOffset<2, 0, O...>::value, ok = PickOffset<2, 1, O...>::value;
    if constexpr ((HMASK >> IN) & 1u) {
      static_assert(oj >= -R1 && oj <= R1 && ok >= -R2 && ok <= R2, "access outside the declared footprint");
      static_assert(FP::BOX || (oj != 0) + (ok != 0) <= 1, "star footprint declared but a diagonal access is used");
      constexpr int h = halo_slot(HMASK, IN), q = e + ok;
      constexpr int vo = q >= 0 ? q / VK : -((-q + VK - 1) / VK);
      const V vec = *reinterpret_cast<const V*>(lp + h * WIN + (r + oj) * LROW + vo * VK);
      return vec[q - vo * VK];
    } else {
      static_assert(oj == 0 && ok == 0, "only halo inputs may be read at an offset");
      return pt[IN][r][e];
    }
  }
  template <int D>
  __device__ __forceinline__ int64_t idx() const {
    static_assert(D >= 0 && D < 2, "index argument out of range");
    return D == 0 ? lj : lk;
  }
};

template <class Body, class T, int NIN, class FP, class TL>
__global__ __launch_bounds__(kWave* TL::WJ* TL::WK) void neptune_apply_tile2(MarchParams<T, NIN> P, Body body) {
  constexpr int RJ = TL::RJ, WJ = TL::WJ, WK = TL::WK, NW = WJ * WK;
  constexpr bool NT = TL::NT;
  using V = typename Vec16<T>::type;
  constexpr int VK = 16 / sizeof(T);
  constexpr int R1 = FP::R0, R2 = FP::R2;
  constexpr unsigned HMASK = FP::HALO_MASK;
  constexpr int NH = popcount_u(HMASK);
  static_assert(NH >= 1, "tile kernel: at least one input read at offsets");
  constexpr int HK = (R2 + VK - 1) / VK * VK;
  constexpr int TJ = WJ * RJ, SPAN = kWave * VK, TK = WK * SPAN;
  constexpr int LROW = TK + 2 * HK, LR = TJ + 2 * R1, WIN = LR * LROW;
  constexpr int NU = 2 * R1 * WK, NUA = NH * NU;
  constexpr int NHW = NUA ? (NUA + NW - 1) / NW : 0, NHWX = NHW ? NHW : 1;
  static_assert(2 * HK <= kWave, "K halo cells are loaded by the first 2*HK lanes");
  static_assert(NH * WIN * (int)sizeof(T) <= 160 * 1024, "windows do not fit the LDS");
  __shared__ __attribute__((aligned(16))) T lds[NH][LR][LROW];

  const int lane = threadIdx.x & (kWave - 1);
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wj = w / WK, wk = w % WK;
  const uint32_t v = xcd_remap(blockIdx.x, gridDim.x);
  const uint32_t kt = v % P.nK, jt = v / P.nK;
  const int32_t j0w = P.rJ0 + (int32_t)(jt * TJ);
  const int32_t j0 = j0w + wj * RJ;
  const int32_t kw0 = (int32_t)(kt * (uint32_t)TK);
  const int32_t kw = kw0 + wk * SPAN;
  const int32_t k0 = kw + lane * VK;
  const bool lane_ok = k0 < P.Ks;
  const uint32_t lane_b = (uint32_t)(k0 < P.Kl ? k0 : P.Kl) * (uint32_t)sizeof(T);

  // rows / cells are addressed per input (InView, apply_march.hpp: inputs 1.. may live in boxes of their own)
  const bool c_left = lane < HK, c_right = lane >= HK && lane < 2 * HK;
  const int32_t cell_k = c_left ? kw0 - HK + lane : kw0 + TK + (lane - HK);
  const int32_t cell_col = c_left ? lane : HK + TK + (lane - HK);
  const bool kh_any = R2 > 0 && ((wk == 0 && c_left) || (wk == WK - 1 && c_right));
  const int32_t own_cell = (R1 + wj * RJ) * LROW + HK + wk * SPAN + lane * VK;
  T* buf = &lds[0][0][0];

  // ---- load the window: my own rows and the halo cells beside them (every input read at offsets), my halo-row units with
  // their corner cells; everything is requested before the first LDS write
  V own[NH][RJ];
  T kh[NH][RJ];
  V jh[NHWX];
  T hc[NHWX];
  V pt[NIN][RJ];
  int32_t hdst[NHWX], hcdst[NHWX], jfixd[NHWX];
  bool hc_any[NHWX], jfix[NHWX];
  static_for<NH>([&](auto hcn) {
    constexpr int h = hcn;
    constexpr int n = halo_input_of(HMASK, h);
    const char* base = reinterpret_cast<const char*>(P.in[n]);
    const uint32_t cell_b = view_cell_bytes(P, n, cell_k);
    static_for<RJ>([&](auto rc) {
      constexpr int r = rc;
      const uint32_t rb = view_row_bytes(P, n, j0 + r);
      own[h][r] = *reinterpret_cast<const V*>(base + (rb + view_lane_bytes(P, n, k0)));
      if constexpr (R2 > 0) { if (kh_any) kh[h][r] = *reinterpret_cast<const T*>(base + (rb + cell_b)); }
    });
  });
  static_for<NHW>([&](auto tc) {
    constexpr int tt = tc;
    const int uu = w + tt * NW;
    const int hh = NU ? uu / NU : 0, u = NU ? uu % NU : 0;
    const int x = u / WK, sp = u % WK;
    const int32_t kc = kw0 + sp * SPAN + lane * VK;
    int n = halo_input_of(HMASK, 0);
    static_for<NH>([&](auto hcn) { constexpr int h = hcn; if (hh == h) n = halo_input_of(HMASK, h); });
    const uint32_t rb = view_row_bytes(P, n, x < R1 ? j0w - R1 + x : j0w + TJ + (x - R1));
    const int32_t lrow = hh * WIN + (x < R1 ? x : TJ + x) * LROW;
    hdst[tt] = uu < NUA ? lrow + HK + sp * SPAN + lane * VK : -1;
    hcdst[tt] = lrow + cell_col;
    hc_any[tt] = FP::BOX && R2 > 0 && uu < NUA && ((sp == 0 && c_left) || (sp == WK - 1 && c_right));
    const char* base = reinterpret_cast<const char*>(P.in[halo_input_of(HMASK, 0)]);
    static_for<NH>([&](auto hcn) { constexpr int h = hcn; if (hh == h) base = reinterpret_cast<const char*>(P.in[halo_input_of(HMASK, h)]); });
    if (hdst[tt] >= 0) jh[tt] = *reinterpret_cast<const V*>(base + (rb + view_lane_bytes(P, n, kc)));
    jfix[tt] = view_fix_star(P, n, kc);
    jfixd[tt] = view_fix_d(P, n);
    if (hc_any[tt]) hc[tt] = *reinterpret_cast<const T*>(base + (rb + view_cell_bytes(P, n, cell_k)));
  });
  static_for<NIN>([&](auto nc) {
    constexpr int n = nc;
    if constexpr (!((HMASK >> n) & 1u)) {
      const char* base = reinterpret_cast<const char*>(P.in[n]);
      static_for<RJ>([&](auto rc) { constexpr int r = rc; pt[n][r] = *reinterpret_cast<const V*>(base + (view_row_bytes(P, n, j0 + r) + view_lane_bytes(P, n, k0))); });
    }
  });

  // ---- lay it out, one barrier, compute
  static_for<NH>([&](auto hcn) {
    constexpr int h = hcn;
    static_for<RJ>([&](auto rc) {
      constexpr int r = rc;
      *reinterpret_cast<V*>(buf + h * WIN + own_cell + r * LROW) =
          view_fix<T>(own[h][r], view_fix_star(P, halo_input_of(HMASK, h), k0), view_fix_d(P, halo_input_of(HMASK, h)));
      if constexpr (R2 > 0) { if (kh_any) buf[h * WIN + (R1 + wj * RJ + r) * LROW + cell_col] = kh[h][r]; }
    });
  });
  static_for<NHW>([&](auto tc) {
    constexpr int tt = tc;
    if (hdst[tt] >= 0) *reinterpret_cast<V*>(buf + hdst[tt]) = view_fix<T>(jh[tt], jfix[tt], jfixd[tt]);
    if (hc_any[tt]) buf[hcdst[tt]] = hc[tt];
  });
  __syncthreads();

  const T* lp = buf + own_cell;
  char* obase = reinterpret_cast<char*>(P.out);
  static_for<RJ>([&](auto rc) {
    constexpr int r = rc;
    const int32_t j = j0 + r;
    const int64_t lj = (int64_t)j + P.olb[1];
    const bool in_j = j >= P.plb[1] && j < P.pub[1];
    V res;
    static_for<VK>([&](auto ec) {
      constexpr int e = ec;
      const int64_t lk = (int64_t)(k0 + e) + P.olb[2];
      const bool inside = in_j && (k0 + e) >= P.plb[2] && (k0 + e) < P.pub[2];
      Tile2Acc<T, NIN, FP, RJ, r, e, LROW, WIN> acc{pt, lp, lj, lk};
      const T val = body(acc);
      T through;
      if constexpr (HMASK & 1u) through = own[0][r][e];   // input 0 owns window 0
      else through = pt[0][r][e];
      res[e] = inside ? val : OutsideOf<Body, T>::apply(body, through);
    });
    if (j < P.rJ1 && lane_ok) {
      V* dst = reinterpret_cast<V*>(obase + (view_row_bytes(P, 0, j) + lane_b));
      if constexpr (NT) __builtin_nontemporal_store(res, dst);
      else *dst = res;
    }
    __builtin_amdgcn_sched_barrier(0);
  });
}

}  // namespace neptune_hip

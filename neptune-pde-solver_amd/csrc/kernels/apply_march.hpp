// apply_march.hpp -- the MI355X stencil kernel for `neptune_ir.apply`: wave tiles that march
// along the slowest axis with the neighbourhood held in registers.
//
// What the reference does for this op (lib/Passes/DataflowLowering.cpp:258-448): malloc the
// result, memcpy input 0 into it (copy-through), then a scalar loop nest that reloads every
// neighbour from memory.  Here one kernel does all of it in a single pass over HBM:
//
//   * layout   : fields stay dense row-major (last dim contiguous, :41-49).  A lane owns VK
//                = 16 B of consecutive cells, a wave owns 64*VK cells of RJ consecutive rows,
//                so every global access is a full 1 KiB coalesced wave transaction.
//   * marching : a wave walks planes i = ib..ie of its (rows x columns) tile.  Plane i+R0+1
//                is being fetched while plane i is computed; the 2*R0+1 live planes sit in
//                VGPRs, so the i-neighbours cost no memory traffic at all.
//   * J halo   : a workgroup is WJ x WK waves; the 2*R1 rows a wave needs beyond its own RJ are its
//                vertical neighbour's own rows and come through LDS (double-buffered, one barrier
//                per plane step; Tile::LDSJ).  Only the workgroup's outermost rows are loaded
//                twice -- L2 hits when tiles that share rows run on the same XCD (xcd_remap).
//   * K halo   : the left/right neighbour cells come from the adjacent lane through a
//                wave shift (DPP wave_shr/wave_shl, or ds_bpermute); the cells just
//                outside the wave's span have wave-uniform addresses and are fetched with
//                scalar loads (s_load through the constant cache), costing SGPRs, not VGPRs.
//                A K radius beyond one lane vector (high-order stars: R2 up to 2*VK) takes a
//                second shift of the shifted value; the wave's edge lanes are patched from the
//                scalar halo cell of the matching distance.
//   * fusion   : copy-through and the bounds test are folded into the store: cells outside
//                apply.bounds get input 0's value, cells inside get body(...).  Nothing is
//                written twice and no intermediate buffer exists.
//   * inputs   : every input read at non-zero offsets (Footprint::HALO_MASK) has its own ring, K
//                halos and prefetch slots; inputs read only at the centre are plain per-plane loads.
//   * forms    : rank 3 marches along dim 0; rank 2 is one plane tiled in (rows, columns)
//                (Tile::JK2) or marches down the rows; rank 1 is a single row.  Rows that are not a
//                whole number of lane vectors are handled with unaligned accesses (MarchParams::Ks).
//
// Bandwidth accounting: algorithmic traffic is one read of every input cell + one write of
// every result cell ((NIN+1)*N*sizeof(T)); redundant fetches are the workgroup-edge J-halo rows
// and 2*R0 planes per chunk of `chunk` planes (measured: 1.01-1.11x, profiles/r01_variant_traffic_*).
#pragma once
#include "apply_common.hpp"

namespace neptune_hip {

// Inputs 1.. may live in boxes of their own (DataflowLowering.cpp:382-410: every input subtracts its OWN lower bound;
// only input 0 must have the result's shape, :283-287) -- a face-located field of extent N+1 beside a cell-located result,
// a field that carries its ghost layers.  The fast kernels take such an input when its box CONTAINS the result box: the
// kernel keeps walking result-physical coordinates (i, j, k) and reads input n at (i, j, k) + sh through the input's own
// row / plane pitch; coordinates that fall outside what the input holds are clamped into [lo, hi] (such reads only ever
// feed cells outside apply.bounds, whose body value is discarded -- an in-bounds access was validated by the plan).
struct InView {
  int64_t plane_b;        // bytes from one plane of this input to the next
  int32_t row_b;          // bytes from one row to the next
  int32_t sh[3];          // (out_lb - in_lb) per kernel axis: result-physical -> input-physical
  int32_t lo[3], hi[3];   // result-physical coordinates the input holds, inclusive: [-sh, extent - sh - 1]
  // The lane right of the result's last vector may find fewer than a lane vector of cells in this input's row (a field
  // on the K faces: ONE more cell).  Its load is clamped to the last whole vector of the row, so the cells it is asked for
  // sit fix_d elements further up in what it loaded: the lane whose vector starts at fix_k rotates its vector down by
  // fix_d elements when the vector is consumed (view_fix; fix_d == 0: nothing to do).
  int32_t fix_k, fix_d;
};

template <class T, int NIN>
struct MarchParams {
  const T* in[NIN];
  T* out;
  int32_t N0, N1, N2;      // extents along (I,J,K) of the result and of input 0 (other inputs: `view`)
  InView view[NIN];        // read by the kernels only when NIN > 1 (view[0] is always the identity)
  // Rows need not be a whole number of 16-byte vectors.  Ks: cells [0,Ks) of every row are stored by
  // this kernel (a multiple of VK); Kl: the last vector start that may be loaded (N2 rounded down to
  // VK, minus VK).  Aligned rows: Ks = N2, Kl = N2-VK.  Ragged rows (N2 % VK != 0): Ks = Kl, i.e. the
  // lane at Ks loads real cells [Ks,Ks+VK) only to serve as its left neighbour's K halo, and the cells
  // [Ks,N2) of every row -- fewer than 2*VK -- are left to a direct-kernel launch by the caller.  A K
  // radius beyond one vector (R2 > VK) reads two lanes to the right: Ks = Kl - VK, so that both are real.
  // Rows then start at any multiple of sizeof(T): loads and stores are unaligned 16-byte accesses.
  int32_t Ks, Kl;
  int32_t plb[3], pub[3];  // apply.bounds in result-physical coordinates (lb - out_lb)
  int64_t olb[3];          // result logical origin (only feeds the region's index arguments)
  int32_t rI0, rI1;        // planes this launch is responsible for (result-physical)
  int32_t rJ0, rJ1;        // rows this launch stores (tiles start at rJ0; the rank-2 tile form's row range)
  int32_t chunk;           // planes per workgroup
  uint32_t nJ, nK;         // tiles along J and K (tiles along I = gridDim.x / (nJ*nK))
};

// view-aware addressing shared by the LDS kernels (apply_plane.hpp): row j / cell k / plane ip are result-physical, input n
// may be a run-time (wave-uniform) index
template <class T, int NIN>
__device__ __forceinline__ uint32_t view_row_bytes(const MarchParams<T, NIN>& P, int n, int32_t j) {
  if (NIN > 1 && n > 0) {
    const InView& vw = P.view[n];
    j = j < vw.lo[1] ? vw.lo[1] : (j > vw.hi[1] ? vw.hi[1] : j);
    return (uint32_t)(j + vw.sh[1]) * (uint32_t)vw.row_b + (uint32_t)vw.sh[2] * (uint32_t)sizeof(T);   // K shift folded in
  }
  j = j < 0 ? 0 : (j >= P.N1 ? P.N1 - 1 : j);
  return (uint32_t)j * (uint32_t)P.N2 * (uint32_t)sizeof(T);
}
template <class T, int NIN>
__device__ __forceinline__ uint32_t view_cell_bytes(const MarchParams<T, NIN>& P, int n, int32_t k) {
  if (NIN > 1 && n > 0) {
    const InView& vw = P.view[n];
    k = k < vw.lo[2] ? vw.lo[2] : (k > vw.hi[2] ? vw.hi[2] : k);
  } else {
    k = k < 0 ? 0 : (k >= P.N2 ? P.N2 - 1 : k);
  }
  return (uint32_t)k * (uint32_t)sizeof(T);
}
// byte offset within a row of the 16-byte vector that starts at cell k (result-physical), clamped so that the whole vector
// lies in the row: the result's rows end at Kl + VK, an input in a box of its own may hold more cells to the right
template <class T, int NIN>
__device__ __forceinline__ uint32_t view_lane_bytes(const MarchParams<T, NIN>& P, int n, int32_t k) {
  constexpr int VK = 16 / (int)sizeof(T);
  int32_t kmax = P.Kl;
  if (NIN > 1 && n > 0) kmax = P.view[n].hi[2] - VK + 1;
  return (uint32_t)(k < kmax ? k : kmax) * (uint32_t)sizeof(T);
}
template <class T, int NIN>
__device__ __forceinline__ const char* view_plane_base(const MarchParams<T, NIN>& P, int n, int32_t ip) {
  if (NIN > 1 && n > 0) {
    const InView& vw = P.view[n];
    const int32_t ic = ip < vw.lo[0] ? vw.lo[0] : (ip > vw.hi[0] ? vw.hi[0] : ip);
    return reinterpret_cast<const char*>(P.in[n]) + (int64_t)(ic + vw.sh[0]) * vw.plane_b;
  }
  const int32_t ic = ip < 0 ? 0 : (ip >= P.N0 ? P.N0 - 1 : ip);
  return reinterpret_cast<const char*>(P.in[n]) + (int64_t)ic * ((int64_t)P.N1 * P.N2 * (int64_t)sizeof(T));
}

template <class T> struct Vec16;
template <> struct Vec16<double> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct Vec16<float> { typedef float type __attribute__((ext_vector_type(4))); };

// InView::fix_k / fix_d: `star` = this lane's vector starts at fix_k (per lane), d = fix_d (wave-uniform, 0 = no fix)
template <class T>
__device__ __forceinline__ typename Vec16<T>::type view_fix(typename Vec16<T>::type v, bool star, int32_t d) {
  if (d != 0) {
    if constexpr (sizeof(T) == 8) {
      v[0] = star ? v[1] : v[0];                       // VK = 2: d can only be 1
    } else {
      if (d & 1) { const bool c = star; typename Vec16<T>::type w = v; v[0] = c ? w[1] : w[0]; v[1] = c ? w[2] : w[1]; v[2] = c ? w[3] : w[2]; }
      if (d & 2) { const bool c = star; typename Vec16<T>::type w = v; v[0] = c ? w[2] : w[0]; v[1] = c ? w[3] : w[1]; }
    }
  }
  return v;
}
template <class T, int NIN>
__device__ __forceinline__ int32_t view_fix_d(const MarchParams<T, NIN>& P, int n) { return (NIN > 1 && n > 0) ? P.view[n].fix_d : 0; }
template <class T, int NIN>
__device__ __forceinline__ bool view_fix_star(const MarchParams<T, NIN>& P, int n, int32_t k) { return NIN > 1 && n > 0 && k == P.view[n].fix_k; }

// ---- wave shifts ----------------------------------------------------------------------
// from_prev: lane l receives x of lane l-1, lane 0 receives `edge`.
// from_next: lane l receives x of lane l+1, lane 63 receives `edge`.
template <bool DPP>
__device__ __forceinline__ int shift32_from_prev(int x, int edge, int lane) {
  if constexpr (DPP) {
    // v_mov_b32_dpp wave_shr:1 ; lane 0 has no source and keeps `old` (= edge)
    return __builtin_amdgcn_update_dpp(edge, x, 0x138, 0xf, 0xf, false);
  } else {
    const int s = __shfl_up(x, 1);
    return lane == 0 ? edge : s;
  }
}
template <bool DPP>
__device__ __forceinline__ int shift32_from_next(int x, int edge, int lane) {
  if constexpr (DPP) {
    // v_mov_b32_dpp wave_shl:1 ; lane 63 keeps `old`
    return __builtin_amdgcn_update_dpp(edge, x, 0x130, 0xf, 0xf, false);
  } else {
    const int s = __shfl_down(x, 1);
    return lane == kWave - 1 ? edge : s;
  }
}
template <bool DPP>
__device__ __forceinline__ float from_prev(float x, float edge, int lane) {
  return __int_as_float(shift32_from_prev<DPP>(__float_as_int(x), __float_as_int(edge), lane));
}
template <bool DPP>
__device__ __forceinline__ float from_next(float x, float edge, int lane) {
  return __int_as_float(shift32_from_next<DPP>(__float_as_int(x), __float_as_int(edge), lane));
}
template <bool DPP>
__device__ __forceinline__ double from_prev(double x, double edge, int lane) {
  const int lo = shift32_from_prev<DPP>(__double2loint(x), __double2loint(edge), lane);
  const int hi = shift32_from_prev<DPP>(__double2hiint(x), __double2hiint(edge), lane);
  return __hiloint2double(hi, lo);
}
template <bool DPP>
__device__ __forceinline__ double from_next(double x, double edge, int lane) {
  const int lo = shift32_from_next<DPP>(__double2loint(x), __double2loint(edge), lane);
  const int hi = shift32_from_next<DPP>(__double2hiint(x), __double2hiint(edge), lane);
  return __hiloint2double(hi, lo);
}

// Wave-uniform read through the scalar data cache (s_load_*).  Only used on kernel inputs,
// which no wave of this launch writes.
template <class T>
__device__ __forceinline__ T scalar_load(const T* p) {
  typedef const __attribute__((address_space(4))) T* cptr;
  return *(cptr)(p);
}

// ---- accessor handed to the body for the cell (row r, element e) of the wave tile -----
// JK (rank 2 only): the field is treated as ONE plane of rows x columns -- (d0,d1) -> (J,K) --
// instead of marching down the rows -- (d0,d1) -> (I,K).
template <class T, int RANK, int NIN, class FP, int RJ, int r, int e, bool JK = false>
struct MarchAcc {
  static constexpr int VK = 16 / sizeof(T);
  static constexpr int R0 = JK ? 0 : FP::R0, R1 = JK ? FP::R0 : FP::R1, R2 = FP::R2;
  static constexpr unsigned HMASK = FP::HALO_MASK;
  static constexpr int NH = popcount_u(HMASK), NHX = NH ? NH : 1;
  static constexpr int NP = 2 * R0 + 1, NR = RJ + 2 * R1, NS = R2 ? R2 : 1;
  // K neighbours exist for every live plane (box) or for the centre plane only (star)
  static constexpr int NPH = FP::BOX ? NP : 1;
  using V = typename Vec16<T>::type;

  const V (&ring)[NHX][NP][NR];
  const T (&lft)[NHX][NPH][NR][NS];
  const T (&rgt)[NHX][NPH][NR][NS];
  const V (&pt)[NIN][RJ];
  int64_t li, lj, lk;  // logical coordinates along (I,J,K)

  template <int IN, int... O>
  __device__ __forceinline__ T get() const {
    static_assert(IN >= 0 && IN < NIN, "input index out of range");
    constexpr int oi = JK ? 0 : PickOffset<RANK, AxisMap<RANK>::I, O...>::value;
    constexpr int oj = JK ? PickOffset<RANK, 0, O...>::value : PickOffset<RANK, AxisMap<RANK>::J, O...>::value;
    constexpr int ok = PickOffset<RANK, AxisMap<RANK>::K, O...>::value;
    if constexpr ((HMASK >> IN) & 1u) {
      constexpr int h = halo_slot(HMASK, IN);
      static_assert(oi >= -R0 && oi <= R0 && oj >= -R1 && oj <= R1 && ok >= -R2 && ok <= R2,
                    "access outside the declared footprint");
      static_assert(FP::BOX || ((oi != 0) + (oj != 0) + (ok != 0) <= 1),
                    "star footprint declared but a diagonal access is used");
      constexpr int p = oi + R0, s = r + R1 + oj, ke = e + ok;
      constexpr int ph = FP::BOX ? p : 0;
      if constexpr (ke < 0) return lft[h][ph][s][R2 + ke];
      else if constexpr (ke >= VK) return rgt[h][ph][s][ke - VK];
      else return ring[h][p][s][ke];
    } else {
      static_assert(oi == 0 && oj == 0 && ok == 0, "only halo inputs may be read at an offset");
      return pt[IN][r][e];
    }
  }
  template <int D>
  __device__ __forceinline__ int64_t idx() const {
    static_assert(D >= 0 && D < RANK, "index argument out of range");
    if constexpr (RANK == 3) return D == 0 ? li : (D == 1 ? lj : lk);
    else if constexpr (RANK == 2) return D == 0 ? (JK ? lj : li) : lk;
    else return lk;
  }
};

// ---- the kernel ---------------------------------------------------------------------------
// Tile: compile-time shape of one workgroup's work
//   RJ    rows per lane            WJ x WK  waves per workgroup along J x K
//   DPP   wave shifts by DPP (true) or ds_bpermute (false)
//   NT    non-temporal stores for the result (it is never re-read by this launch)
//   PF    prefetch distance: planes of row loads kept in flight per wave.  The kernel is
//         bound by bytes in flight per CU (HBM latency under load is several microseconds), so
//         PF trades VGPRs (occupancy) for deeper per-wave queues.
//   NTL   non-temporal loads for the input rows
//   LDSJ  J-halo rows come from the neighbouring wave of the same workgroup through LDS
//         instead of a second global load.  Two waves that miss on the same line at almost the
//         same time are BOTH served from the fabric (measured: every halo-row load of the
//         all-global form reached the memory side, FETCH = 1.5x the field for RJ = 4), so
//         sharing through LDS removes real HBM/fabric traffic, not just L2 hits.  Only the
//         workgroup's outermost rows still come from global memory.
//   JK2   rank 2 only: no marching; the field is one plane tiled in (rows, columns).  Waves are
//         short-lived (load RJ rows + exchanged halo rows, compute, store, exit), like the fastest
//         plain copy kernel; the J halo goes through LDS.
//   JHL   star stencils: obtain the J halo rows when a plane becomes the centre (see JH_LATE in
//         the kernel) instead of when it arrives
//   KD    star stencils: how many plane steps ahead of their use the scalar K halos are requested (1 .. 3).  The
//         cell just outside a wave's span is the edge cell of the K-neighbouring workgroup's row: requested in the
//         same step as the rows of the same plane (KD = R0 + PF) the two requests for one line fall close together in
//         time whichever workgroup is ahead, and the later one finds the line in the XCD's L2; requested a step later
//         (KD = 1) the line has often been evicted again (an XCD streams ~5 MiB per plane step through its 4 MiB L2).
//   PLN   run the plane-in-LDS kernel (apply_plane.hpp) instead of this one: rank-3 stars, J and K neighbours read from the
//         centre plane laid out in LDS; only RJ, WJ, WK, PF and NT mean anything there
template <int RJ_, int WJ_, int WK_, bool DPP_, bool NT_, int PF_, bool NTL_, bool LDSJ_ = false, bool JK2_ = false,
          bool JHL_ = false, int KD_ = 1, bool PLN_ = false>
struct Tile {
  static constexpr int RJ = RJ_, WJ = WJ_, WK = WK_, PF = PF_, KD = KD_;
  static constexpr bool DPP = DPP_, NT = NT_, NTL = NTL_, LDSJ = LDSJ_, JK2 = JK2_, JHL = JHL_, PLN = PLN_;
};

template <class Body, class T, int RANK, int NIN, class FP, class TL>
__global__ __launch_bounds__(kWave* TL::WJ* TL::WK) void neptune_apply_march(MarchParams<T, NIN> P, Body body) {
  constexpr int RJ = TL::RJ, WJ = TL::WJ, WK = TL::WK, PF = TL::PF;
  constexpr bool DPP = TL::DPP, NT = TL::NT, NTL = TL::NTL;
  static_assert(PF >= 1, "prefetch distance");
  static_assert(FP::R2 <= 2 * (16 / (int)sizeof(T)), "K neighbours come from at most two lanes away");
  using V = typename Vec16<T>::type;
  constexpr int VK = 16 / sizeof(T);
  constexpr bool JK = RANK == 2 && TL::JK2;
  constexpr int R0 = JK ? 0 : FP::R0, R1 = JK ? FP::R0 : FP::R1, R2 = FP::R2;
  constexpr unsigned HMASK = FP::HALO_MASK;
  constexpr int NH = popcount_u(HMASK), NHX = NH ? NH : 1;  // halo inputs, each with its own ring
  constexpr bool BOX = FP::BOX;
  constexpr bool HAS_HALO = NH > 0;
  constexpr int NP = 2 * R0 + 1, NR = RJ + 2 * R1, NS = R2 ? R2 : 1;
  // Scalar K halos: a box stencil needs them for every live plane, so they travel with the
  // plane through the ring (fetched R0+1 steps ahead).  A star stencil needs them for the
  // centre plane only, so they are fetched one step ahead and never ride the ring: 1/3 fewer
  // live SGPRs for the 7-point stencil.
  constexpr int NPH = BOX ? NP : 1;
  constexpr int HLEAD = BOX ? R0 : 0;  // halos in flight belong to plane i+1+HLEAD
  // star stencils may request their K halos KD > 1 steps ahead (Tile::KD); a box stencil's ride the ring
  constexpr int KD = (BOX || !HAS_HALO || R2 == 0 || JK) ? 1 : TL::KD;
  static_assert(KD >= 1 && KD <= 3, "K halo lead");
  constexpr int U = (PF % KD == 0) ? PF : (KD % PF == 0 ? KD : PF * KD);  // steps per loop trip: slots stay compile-time
  // J halo rows follow the same split.  A box stencil reads them on every live plane: they are
  // loaded / exchanged when a plane arrives and ride the ring.  A star stencil reads them on the
  // centre plane only: they are obtained when a plane BECOMES the centre (LDS exchange of the
  // centre's own rows; workgroup-edge rows fetched PF steps ahead -- one step ahead measured 10 %
  // slower, the edge waves then stall the whole workgroup at the barrier), so no halo row rides the
  // ring for the planes between arrival and centre.
  constexpr bool JH_LATE = TL::JHL && !BOX && R1 > 0;
  constexpr int PJ = JH_LATE ? R0 : NP - 1;  // plane whose J halo rows are filled at a step
  constexpr int NJH = R1 > 0 ? 2 * R1 : 1;
  static_assert(HAS_HALO || (R0 == 0 && R1 == 0 && R2 == 0), "pointwise footprint must have zero radii");
  // J-halo exchange through LDS needs a vertical neighbour in the workgroup and a J radius that
  // one neighbour can serve
  // ... and the exchange buffers (2 x halo inputs x waves x 2*R1 rows x 1 KiB) must leave room for a second
  // workgroup per CU (160 KiB LDS): beyond 64 KiB -- three halo inputs of radius 2 on an 8-wave tile -- the
  // halo rows come from global memory like in the tiles without LDSJ
  // A J radius larger than the rows per lane (radius 3-4 stars on the 2-row tiles) reaches past the adjacent wave: every wave
  // then publishes ALL its own rows and takes halo row x from the wave ceil(distance / RJ) away (LDS_ALL); rows beyond
  // the workgroup's window still come from global memory, loaded by the waves close enough to the window's edge.
  constexpr bool LDS_ALL = R1 > RJ;
  constexpr int LPUB = LDS_ALL ? RJ : 2 * R1;   // rows a wave publishes per halo input and step
  constexpr bool LDSJ = TL::LDSJ && WJ > 1 && R1 > 0 && HAS_HALO &&
                        2 * NHX * WJ * WK * LPUB * kWave * (int)sizeof(V) <= 64 * 1024;
  constexpr int LROWS = LDSJ ? LPUB : 1;
  // how many waves away halo row x lives (x < R1: rows above the tile, nearest last; x >= R1: rows below, nearest first)
  // and which of that wave's own rows it is
  auto halo_dist = [](int x) constexpr { return x < R1 ? (R1 - x + RJ - 1) / RJ : (x - R1) / RJ + 1; };
  auto halo_row = [](int x) constexpr { return x < R1 ? RJ * ((R1 - x + RJ - 1) / RJ) - (R1 - x) : (x - R1) % RJ; };
  // [double buffer][halo input][wave][published rows][lane]: 1 KiB per row.  Adjacent-wave form: first R1 own rows | last R1
  __shared__ V lds_rows[LDSJ ? 2 : 1][LDSJ ? NHX : 1][LDSJ ? WJ * WK : 1][LROWS][LDSJ ? kWave : 1];

  const int lane = threadIdx.x & (kWave - 1);
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wj = w / WK, wk = w % WK;

  // tile decode: K tiles fastest, then J tiles, then chunks of planes
  // (tried and measured slower: J tiles fastest, chunks fastest, and no XCD remap -- profiles/r02_headline_search.txt)
  const uint32_t v = xcd_remap(blockIdx.x, gridDim.x);
  const uint32_t kt = v % P.nK;
  const uint32_t t = v / P.nK;
  const uint32_t jt = t % P.nJ;
  const uint32_t ct = t / P.nJ;

  // All in-plane index math is 32-bit (the host guarantees a plane is < 2 GiB); only the plane
  // base is a 64-bit pointer.  Wave-uniform values live in SGPRs.
  const int32_t j0 = P.rJ0 + (int32_t)(jt * (WJ * RJ)) + wj * RJ;      // first own row
  const int32_t kw = (int32_t)((kt * WK + wk) * (uint32_t)(kWave * VK));  // first own column
  // Waves whose tile lies outside the field are not retired: every address below is clamped
  // into the field and their stores are predicated off, so they can keep taking part in the
  // workgroup barriers of the LDS exchange.
  const int32_t k0 = kw + lane * VK;
  const bool lane_ok = k0 < P.Ks;  // Ks % VK == 0, so a lane is entirely in or out
  const uint32_t lane_b = (uint32_t)(k0 < P.Kl ? k0 : P.Kl) * (uint32_t)sizeof(T);
  const int32_t kw_end = kw + kWave * VK;
  // ... per input when there are several: an input in a box of its own may hold whole vectors right of the result's rows
  uint32_t lane_bv[NIN > 1 ? NIN : 1];
  int32_t fixd[NIN > 1 ? NIN : 1];     // InView::fix_d per input (wave-uniform) and whether this lane is the one to fix
  bool fixs[NIN > 1 ? NIN : 1];
  static_for<(NIN > 1 ? NIN : 1)>([&](auto nc) {
    constexpr int n = nc;
    lane_bv[n] = view_lane_bytes(P, n, k0);
    fixd[n] = view_fix_d(P, n);
    fixs[n] = view_fix_star(P, n, k0);
  });

  const int32_t ib = P.rI0 + (int32_t)ct * P.chunk;
  const int32_t ie = (ib + P.chunk < P.rI1) ? ib + P.chunk : P.rI1;
  if (ib >= ie) return;
  const int64_t plane_b = (int64_t)P.N1 * P.N2 * (int64_t)sizeof(T);

  // byte offsets of the rows this wave touches (halo rows clamped into the field: a clamped
  // row is only ever read for cells outside apply.bounds, whose body value is discarded).  One set for the result and
  // input 0; with several inputs one set per input (InView: own pitch, own clamp range, K shift folded in).
  constexpr int NV = NIN > 1 ? NIN : 1;
  uint32_t rowb[NV][NR];
  static_for<NV>([&](auto nc) {
    constexpr int n = nc;
    static_for<NR>([&](auto sc) {
      constexpr int s = sc;
      int32_t j = j0 + (s - R1);
      if constexpr (NIN > 1 && n > 0) {
        const InView& vw = P.view[n];
        j = j < vw.lo[1] ? vw.lo[1] : (j > vw.hi[1] ? vw.hi[1] : j);
        rowb[n][s] = (uint32_t)(j + vw.sh[1]) * (uint32_t)vw.row_b + (uint32_t)vw.sh[2] * (uint32_t)sizeof(T);
      } else {
        j = j < 0 ? 0 : (j >= P.N1 ? P.N1 - 1 : j);
        rowb[n][s] = (uint32_t)j * (uint32_t)P.N2 * (uint32_t)sizeof(T);
      }
    });
  });
  // byte offsets (within a row) of the cells just outside the wave's span, clamped
  uint32_t khlb[NV][NS], khrb[NV][NS];
  static_for<NV>([&](auto nc) {
    constexpr int n = nc;
    static_for<NS>([&](auto xc) {
      constexpr int x = xc;
      int32_t kl = kw - R2 + x, kr = kw_end + x;
      if constexpr (NIN > 1 && n > 0) {
        const InView& vw = P.view[n];
        kl = kl < vw.lo[2] ? vw.lo[2] : (kl > vw.hi[2] ? vw.hi[2] : kl);
        kr = kr < vw.lo[2] ? vw.lo[2] : (kr > vw.hi[2] ? vw.hi[2] : kr);
      } else {
        kl = kl < 0 ? 0 : (kl >= P.N2 ? P.N2 - 1 : kl);
        kr = kr >= P.N2 ? P.N2 - 1 : kr;
      }
      khlb[n][x] = (uint32_t)kl * (uint32_t)sizeof(T);
      khrb[n][x] = (uint32_t)kr * (uint32_t)sizeof(T);
    });
  });

  // ---- register state ----
  V ring[NHX][NP][NR];      // live planes of each halo input (row vectors)
  T khl[NHX][NPH][NR][NS];  // scalar K halos, left  (wave-uniform)
  T khr[NHX][NPH][NR][NS];  // scalar K halos, right
  V nxt[NHX][PF][NR];       // planes in flight (slot ph is consumed by steps i == ph mod PF)
  V njh[NHX][PF][NJH];      // star stencils: J halo rows of the next PF centre planes, in flight
  T nkhl[NHX][KD][NR][NS], nkhr[NHX][KD][NR][NS];  // K halos in flight: slot d is consumed by steps i == d mod KD
  V pt[NIN][RJ];            // inputs read at offset 0 only, current plane
  V npt[NIN][RJ];           // ... next plane, in flight

  // rows whose K halo is needed: own rows always, J-halo rows only for box stencils
  auto need_khalo = [](int s) constexpr { return R2 > 0 && (BOX || (s >= R1 && s < R1 + RJ)); };

  // plane ip (result-physical, clamped into what the input holds) of input n
  auto plane_base = [&](auto nc, int32_t ip) -> const char* {
    constexpr int n = decltype(nc)::value;
    if constexpr (NIN > 1 && n > 0) {
      const InView& vw = P.view[n];
      const int32_t ic = ip < vw.lo[0] ? vw.lo[0] : (ip > vw.hi[0] ? vw.hi[0] : ip);
      return reinterpret_cast<const char*>(P.in[n]) + (int64_t)(ic + vw.sh[0]) * vw.plane_b;
    } else {
      const int32_t ic = ip < 0 ? 0 : (ip >= P.N0 ? P.N0 - 1 : ip);
      return reinterpret_cast<const char*>(P.in[n]) + (int64_t)ic * plane_b;
    }
  };
  constexpr auto vidx = [](int n) constexpr { return NIN > 1 ? n : 0; };   // which row / halo offset set input n uses
  // all_rows = true: fetch the J-halo rows from global memory too (prologue planes, which do
  // not pass through the LDS exchange)
  auto load_rows = [&](auto hc, int32_t ip, V(&rows)[NR], auto all_rows_c) {
    constexpr bool all_rows = decltype(all_rows_c)::value;
    if constexpr (HAS_HALO) {
      constexpr int hin = halo_input_of(HMASK, decltype(hc)::value);
      const char* base = plane_base(std::integral_constant<int, hin>{}, ip);
      static_for<NR>([&](auto sc) {
        constexpr int s = sc;
        auto ld = [&] {
          const V* src = reinterpret_cast<const V*>(base + (rowb[vidx(hin)][s] + lane_bv[vidx(hin)]));
          if constexpr (NTL) rows[s] = __builtin_nontemporal_load(src);
          else rows[s] = *src;
        };
        constexpr bool is_halo_row = s < R1 || s >= R1 + RJ;
        if constexpr (JH_LATE && is_halo_row) {
          // star: halo rows are fetched by load_jhalo when the plane becomes the centre
        } else if constexpr (LDSJ && !all_rows && s < R1) {
          if (wj < halo_dist(s)) ld();            // above the workgroup's window: no wave to get them from
        } else if constexpr (LDSJ && !all_rows && s >= R1 + RJ) {
          if (wj + halo_dist(s - RJ) > WJ - 1) ld();   // below the window (halo index of slot s: s - RJ)
        } else {
          ld();
        }
      });
    }
  };
  // star stencils: the 2*R1 J-halo rows of plane ip (x < R1: rows above the tile, else below).  With
  // the LDS exchange only the workgroup's outermost waves need them from global memory.
  auto load_jhalo = [&](auto hc, int32_t ip, V(&jh)[NJH]) {
    if constexpr (JH_LATE && HAS_HALO) {
      constexpr int hin = halo_input_of(HMASK, decltype(hc)::value);
      const char* base = plane_base(std::integral_constant<int, hin>{}, ip);
      static_for<2 * R1>([&](auto xc) {
        constexpr int x = xc;
        constexpr int s = x < R1 ? x : RJ + x;  // ring slot of that row (R1 + RJ + (x - R1))
        const bool want = !LDSJ || (x < R1 ? wj < halo_dist(x) : wj + halo_dist(x) > WJ - 1);
        if (want) {
          const V* src = reinterpret_cast<const V*>(base + (rowb[vidx(hin)][s] + lane_bv[vidx(hin)]));
          if constexpr (NTL) jh[x] = __builtin_nontemporal_load(src);
          else jh[x] = *src;
        }
      });
    }
  };
  auto load_khalos = [&](auto hc, int32_t ip, T(&hl)[NR][NS], T(&hr)[NR][NS]) {
    if constexpr (HAS_HALO && R2 > 0) {
      constexpr int hin = halo_input_of(HMASK, decltype(hc)::value);
      const char* base = plane_base(std::integral_constant<int, hin>{}, ip);
      static_for<NR>([&](auto sc) {
        constexpr int s = sc;
        if constexpr (need_khalo(s)) {
          static_for<R2>([&](auto xc) {
            constexpr int x = xc;
            hl[s][x] = scalar_load(reinterpret_cast<const T*>(base + (rowb[vidx(hin)][s] + khlb[vidx(hin)][x])));
            hr[s][x] = scalar_load(reinterpret_cast<const T*>(base + (rowb[vidx(hin)][s] + khrb[vidx(hin)][x])));
          });
        }
      });
    }
  };
  auto load_point_inputs = [&](int32_t ip, V(&dst)[NIN][RJ]) {
    static_for<NIN>([&](auto nc) {
      constexpr int n = nc;
      if constexpr (!((HMASK >> n) & 1u)) {
        const char* base = plane_base(nc, ip);
        static_for<RJ>([&](auto rc) {
          constexpr int r = rc;
          dst[n][r] = *reinterpret_cast<const V*>(base + (rowb[vidx(n)][r + R1] + lane_bv[vidx(n)]));
        });
      }
    });
  };

  // loop-invariant predicates
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

  // ---- prologue: planes ib-R0 .. ib+R0-1 into ring[1..NP-1]; planes ib+R0 .. ib+R0+PF-1 in flight
  static_for<NH>([&](auto hc) {
    constexpr int h = hc;
    static_for<NP - 1>([&](auto pc) {
      constexpr int p = pc;
      load_rows(hc, ib - R0 + p, ring[h][p + 1], std::true_type{});  // shifted down at the top of the first step
      if constexpr (BOX) load_khalos(hc, ib - R0 + p, khl[h][p + 1], khr[h][p + 1]);
    });
    static_for<PF>([&](auto dc) {
      constexpr int d = dc;
      load_rows(hc, ib + R0 + d, nxt[h][d], std::false_type{});
    });
    static_for<KD>([&](auto dc) {
      constexpr int d = dc;
      load_khalos(hc, ib + HLEAD + d, nkhl[h][d], nkhr[h][d]);
    });
    static_for<PF>([&](auto dc) {
      constexpr int d = dc;
      load_jhalo(hc, ib + d, njh[h][d]);
    });
  });
  load_point_inputs(ib, npt);
  if constexpr (NIN > 1) {
    // the prologue's planes went straight into the ring: the same fix, once
    static_for<NH>([&](auto hc) {
      constexpr int h = hc, hin = halo_input_of(HMASK, h);
      if constexpr (hin > 0) {
        if (fixd[hin] != 0)
          static_for<NP - 1>([&](auto pc) {
            constexpr int p = pc;
            static_for<NR>([&](auto sc) { constexpr int s = sc; ring[h][p + 1][s] = view_fix<T>(ring[h][p + 1][s], fixs[hin], fixd[hin]); });
          });
      }
    });
  }

  // one plane step; `slot` (compile-time) names the in-flight buffer holding plane i+R0, so no
  // register that a load is still writing is ever moved
  auto step = [&](const int32_t i, auto phase_c) {
    constexpr int slot = decltype(phase_c)::value % PF, kslot = decltype(phase_c)::value % KD;
    // rotate: ring[p] <- ring[p+1], newest plane <- nxt[slot]
    static_for<NH>([&](auto hc) {
      constexpr int h = hc;
      static_for<NP - 1>([&](auto pc) {
        constexpr int p = pc;
        static_for<NR>([&](auto sc) {
          constexpr int s = sc;
          if constexpr (!JH_LATE || (s >= R1 && s < R1 + RJ)) ring[h][p][s] = ring[h][p + 1][s];
          if constexpr (BOX) {
            static_for<NS>([&](auto xc) {
              constexpr int x = xc;
              khl[h][p][s][x] = khl[h][p + 1][s][x];
              khr[h][p][s][x] = khr[h][p + 1][s][x];
            });
          }
        });
      });
      constexpr int hin = halo_input_of(HMASK, h);
      static_for<NR>([&](auto sc) {
        constexpr int s = sc;
        if constexpr (!JH_LATE || (s >= R1 && s < R1 + RJ)) {
          if constexpr (NIN > 1 && hin > 0) ring[h][NP - 1][s] = view_fix<T>(nxt[h][slot][s], fixs[hin], fixd[hin]);
          else ring[h][NP - 1][s] = nxt[h][slot][s];
        }
        static_for<NS>([&](auto xc) {
          constexpr int x = xc;
          khl[h][NPH - 1][s][x] = nkhl[h][kslot][s][x];
          khr[h][NPH - 1][s][x] = nkhr[h][kslot][s][x];
        });
      });
      if constexpr (JH_LATE) {
        // star: the centre plane's halo rows, fetched from global memory PF steps ago (by all waves
        // without the LDS exchange, by the workgroup's outermost waves with it)
        static_for<2 * R1>([&](auto xc) {
          constexpr int x = xc;
          if constexpr (NIN > 1 && hin > 0) ring[h][PJ][x < R1 ? x : RJ + x] = view_fix<T>(njh[h][slot][x], fixs[hin], fixd[hin]);
          else ring[h][PJ][x < R1 ? x : RJ + x] = njh[h][slot][x];
        });
      }
    });
    if constexpr (LDSJ) {
      // J-halo rows of plane PJ (box: the plane that just arrived; star: the plane that just became
      // the centre), for every halo input: publish my first/last R1 own rows, take the neighbouring
      // waves' rows.  Double-buffered by step parity, ONE barrier per step.
      const int buf = (i - ib) & 1;
      static_for<NH>([&](auto hc) {
        constexpr int h = hc;
        if constexpr (LDS_ALL) {
          static_for<RJ>([&](auto rc) { constexpr int r = rc; lds_rows[buf][h][w][r][lane] = ring[h][PJ][R1 + r]; });
        } else {
          static_for<R1>([&](auto xc) {
            constexpr int x = xc;
            lds_rows[buf][h][w][x][lane] = ring[h][PJ][R1 + x];            // first own rows
            lds_rows[buf][h][w][R1 + x][lane] = ring[h][PJ][RJ + x];       // last own rows (s = R1+RJ-R1+x)
          });
        }
      });
      __syncthreads();
      static_for<NH>([&](auto hc) {
        constexpr int h = hc;
        if constexpr (LDS_ALL) {
          static_for<2 * R1>([&](auto xc) {
            constexpr int x = xc;                       // halo index: x < R1 above, else below
            constexpr int d = halo_dist(x), r = halo_row(x);
            constexpr int s = x < R1 ? x : RJ + x;      // ring slot of that row
            if constexpr (x < R1) { if (wj >= d) ring[h][PJ][s] = lds_rows[buf][h][w - d * WK][r][lane]; }
            else { if (wj + d <= WJ - 1) ring[h][PJ][s] = lds_rows[buf][h][w + d * WK][r][lane]; }
          });
        } else {
          static_for<R1>([&](auto xc) {
            constexpr int x = xc;
            // rows above my tile = last rows of the wave above; rows below = first rows of the wave below
            if (wj > 0) ring[h][PJ][x] = lds_rows[buf][h][w - WK][R1 + x][lane];
            if (wj < WJ - 1) ring[h][PJ][R1 + RJ + x] = lds_rows[buf][h][w + WK][x][lane];
          });
        }
      });
    }
    static_for<NIN>([&](auto nc) {
      constexpr int n = nc;
      static_for<RJ>([&](auto rc) { constexpr int r = rc; pt[n][r] = npt[n][r]; });
    });
    // refill the slot just consumed with the plane PF steps ahead; K halos and offset-0 inputs
    // are one step ahead
    static_for<NH>([&](auto hc) {
      constexpr int h = hc;
      if (i + PF < ie) {
        load_rows(hc, i + PF + R0, nxt[h][slot], std::false_type{});
        load_jhalo(hc, i + PF, njh[h][slot]);  // the plane that is the centre PF steps from now
      }
      if (i + KD < ie) load_khalos(hc, i + KD + HLEAD, nkhl[h][kslot], nkhr[h][kslot]);
    });
    if (i + 1 < ie) load_point_inputs(i + 1, npt);

    // K neighbours from adjacent lanes (edges from the scalar halos)
    T lft[NHX][NPH][NR][NS], rgt[NHX][NPH][NR][NS];
    if constexpr (R2 > 0) {
      static_for<NH>([&](auto hc) {
        constexpr int h = hc;
        static_for<NPH>([&](auto pc) {
          constexpr int ph = pc;
          constexpr int p = BOX ? ph : R0;  // star: centre plane only
          static_for<NR>([&](auto sc) {
            constexpr int s = sc;
            // star stencils read K neighbours only on own rows
            if constexpr (BOX || (s >= R1 && s < R1 + RJ)) {
              static_for<R2>([&](auto xc) {
                constexpr int x = xc;
                // lft[x] = cell k0 - dl, rgt[x] = cell k0 + VK - 1 + dr.  Up to VK cells away the value sits in the
                // adjacent lane: one wave shift, the wave's edge lane takes the scalar halo cell.  From VK+1 to 2*VK
                // cells away (radius 3-4 in fp64) it sits two lanes away: two shifts -- the first one feeds the
                // outermost lane the nearer halo cell, which the second shift hands to its neighbour while the
                // outermost lane itself receives the farther halo cell.
                constexpr int dl = R2 - x, dr = x + 1;
                if constexpr (dl <= VK) {
                  lft[h][ph][s][x] = from_prev<DPP>(ring[h][p][s][VK - dl], khl[h][ph][s][x], lane);
                } else {
                  const T one = from_prev<DPP>(ring[h][p][s][2 * VK - dl], khl[h][ph][s][R2 - (dl - VK)], lane);
                  lft[h][ph][s][x] = from_prev<DPP>(one, khl[h][ph][s][x], lane);
                }
                if constexpr (dr <= VK) {
                  rgt[h][ph][s][x] = from_next<DPP>(ring[h][p][s][dr - 1], khr[h][ph][s][x], lane);
                } else {
                  const T one = from_next<DPP>(ring[h][p][s][dr - 1 - VK], khr[h][ph][s][dr - 1 - VK], lane);
                  rgt[h][ph][s][x] = from_next<DPP>(one, khr[h][ph][s][x], lane);
                }
              });
            }
          });
        });
      });
    }

    const bool in_i = i >= P.plb[0] && i < P.pub[0];
    const int64_t li = (int64_t)i + P.olb[0];
    char* obase = reinterpret_cast<char*>(P.out) + (int64_t)i * plane_b;
    static_for<RJ>([&](auto rc) {
      constexpr int r = rc;
      const int64_t lj = (int64_t)(j0 + r) + P.olb[1];
      const bool in_ij = in_i && in_j[r];
      V res;
      static_for<VK>([&](auto ec) {
        constexpr int e = ec;
        const int64_t lk = (int64_t)(k0 + e) + P.olb[2];
        const bool inside = in_ij && in_k[e];
        MarchAcc<T, RANK, NIN, FP, RJ, r, e, JK> acc{ring, lft, rgt, pt, li, lj, lk};
        const T val = body(acc);
        T through;  // copy-through source: input 0 at the same physical index
        if constexpr (HMASK & 1u) through = ring[0][R0][r + R1][e];  // input 0 owns ring slot 0
        else through = pt[0][r][e];
        res[e] = inside ? val : OutsideOf<Body, T>::apply(body, through);
      });
      if (row_ok[r] && lane_ok) {
        // rowb[0][r+R1] is this own row's offset in the result (own rows are never clamped when row_ok)
        V* dst = reinterpret_cast<V*>(obase + (rowb[0][r + R1] + lane_b));
        if constexpr (NT) __builtin_nontemporal_store(res, dst);
        else *dst = res;
      }
    });
  };

  for (int32_t i = ib; i < ie; i += U) {
    static_for<U>([&](auto phc) {
      constexpr int ph = phc;
      if (i + ph < ie) step(i + ph, phc);
    });
  }
}

}  // namespace neptune_hip

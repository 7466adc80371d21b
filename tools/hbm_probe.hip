// hbm_probe.hip -- what bounds the march kernel?  Copy kernels with the march kernel's ACCESS PATTERN (wave tiles
// of RJ rows x 1 KiB marching along dim 0 of a dense 3-D field, plane i+PF in flight while plane i is stored) but no
// halo, no LDS and no arithmetic, next to the plain linear copies.  If such a copy runs at the linear copy's rate, the
// stencil's remaining gap is its halo traffic and synchronisation; if it does not, the pattern itself (tile shape,
// chunk length, workgroup size) is the lever.  Measurement tool only: nothing here is part of the product.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hbm_probe.hip -o build/hbm_probe && build/hbm_probe [N0 N1 N2]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(e)                                                                        \
  do {                                                                                  \
    hipError_t _e = (e);                                                                \
    if (_e != hipSuccess) {                                                             \
      fprintf(stderr, "%s failed: %s (%s:%d)\n", #e, hipGetErrorString(_e), __FILE__, __LINE__); \
      exit(1);                                                                          \
    }                                                                                   \
  } while (0)

typedef double V2 __attribute__((ext_vector_type(2)));
constexpr int kWave = 64;

static uint32_t xcd_remap_host(uint32_t b, uint32_t nb) {
  const uint32_t q = nb / 8, r = nb % 8;
  const uint32_t xcd = b % 8, pos = b / 8;
  return xcd < r ? xcd * (q + 1) + pos : r * (q + 1) + (xcd - r) * q + pos;
}
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nb) {
  const uint32_t q = nb / 8, r = nb % 8;
  const uint32_t xcd = b % 8, pos = b / 8;
  return xcd < r ? xcd * (q + 1) + pos : r * (q + 1) + (xcd - r) * q + pos;
}

template <class F, int... Is>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  sfor_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

struct P3 {
  const double* in;
  double* out;
  int32_t N0, N1, N2, chunk;
  uint32_t nJ, nK, nC;
  int32_t order;  // 0: K tiles fastest, then J, then chunks (the march kernel's numbering); 1: J fastest, K, chunks; 2: chunks fastest
  uint64_t* ts;   // timeline mode: per workgroup {start, first store issued, end} on the 100 MHz wall clock (null: off)
};

// the march kernel's traversal without its stencil: SYNC adds the per-plane workgroup barrier, REMAP the XCD-aware
// tile numbering, HALO loads (and discards into the sum) the two J-halo rows of the workgroup's outermost waves
template <int RJ, int WJ, int WK, int PF, bool SYNC, bool REMAP, bool NT, bool NTL = false>
__global__ __launch_bounds__(kWave* WJ* WK) void march_copy(P3 P) {
  extern __shared__ char dyn_lds[];  // only there to limit residency (launch parameter)
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wj = w / WK, wk = w % WK;
  const uint32_t v = REMAP ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  if (P.ts && threadIdx.x == 0) P.ts[3 * blockIdx.x] = wall_clock64();
  uint32_t kt, jt, ct;
  if (P.order == 0) { kt = v % P.nK; const uint32_t t = v / P.nK; jt = t % P.nJ; ct = t / P.nJ; }
  else if (P.order == 1) { jt = v % P.nJ; const uint32_t t = v / P.nJ; kt = t % P.nK; ct = t / P.nK; }
  else { ct = v % P.nC; const uint32_t t = v / P.nC; kt = t % P.nK; jt = t / P.nK; }
  const int32_t j0 = (int32_t)(jt * (WJ * RJ)) + wj * RJ;
  const int32_t k0 = (int32_t)((kt * WK + wk) * 128u) + lane * 2;
  if (j0 >= P.N1 || k0 >= P.N2) {
    if (SYNC) {
      // stay for the barriers
    } else {
      return;
    }
  }
  const int32_t ib = (int32_t)ct * P.chunk;
  const int32_t ie = ib + P.chunk < P.N0 ? ib + P.chunk : P.N0;
  const int64_t plane = (int64_t)P.N1 * P.N2;
  const bool ok = j0 < P.N1 && k0 < P.N2;
  const int64_t base = ok ? (int64_t)j0 * P.N2 + k0 : 0;
  V2 nxt[PF][RJ];
  sfor<PF>([&](auto dc) {
    constexpr int d = dc;
    const int32_t ip = ib + d < P.N0 ? ib + d : P.N0 - 1;
    sfor<RJ>([&](auto rc) {
      constexpr int r = rc;
      const V2* src = reinterpret_cast<const V2*>(P.in + (int64_t)ip * plane + base + (int64_t)r * P.N2);
      nxt[d][r] = NTL ? __builtin_nontemporal_load(src) : *src;
    });
  });
  auto step = [&](int32_t i, auto slot_c) {
    constexpr int slot = slot_c;
    V2 cur[RJ];
    sfor<RJ>([&](auto rc) { constexpr int r = rc; cur[r] = nxt[slot][r]; });
    if (SYNC) __syncthreads();
    if (i + PF < ie) {
      sfor<RJ>([&](auto rc) {
        constexpr int r = rc;
        const V2* src = reinterpret_cast<const V2*>(P.in + (int64_t)(i + PF) * plane + base + (int64_t)r * P.N2);
        nxt[slot][r] = NTL ? __builtin_nontemporal_load(src) : *src;
      });
    }
    if (ok) {
      sfor<RJ>([&](auto rc) {
        constexpr int r = rc;
        V2* dst = reinterpret_cast<V2*>(P.out + (int64_t)i * plane + base + (int64_t)r * P.N2);
        if (NT) __builtin_nontemporal_store(cur[r], dst);
        else *dst = cur[r];
      });
    }
  };
  for (int32_t i = ib; i < ie; i += PF) {
    sfor<PF>([&](auto phc) {
      constexpr int ph = phc;
      if (i + ph < ie) step(i + ph, phc);
    });
    if (P.ts && threadIdx.x == 0 && i == ib) P.ts[3 * blockIdx.x + 1] = wall_clock64();
  }
  if (P.ts && threadIdx.x == 0) {
    __builtin_amdgcn_s_waitcnt(0);   // this wave's stores have left
    P.ts[3 * blockIdx.x + 2] = wall_clock64();
  }
}

// the same traversal (rj4 x wj16 x wk1, pf1, barrier, XCD remap) through BUFFER loads/stores, whose cache-policy bits the
// compiler lets us choose (aux: 1 = sc0, 2 = nt, 16 = sc1) while still tracking the waits
typedef unsigned int U4 __attribute__((ext_vector_type(4)));
template <int LDAUX, int STAUX>
__global__ __launch_bounds__(1024) void march_copy_buf(P3 P) {
  extern __shared__ char dyn_lds[];
  constexpr int RJ = 4, WJ = 16;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t v = xcd_remap(blockIdx.x, gridDim.x);
  const uint32_t kt = v % P.nK, t = v / P.nK, jt = t % P.nJ, ct = t / P.nJ;
  const int32_t j0 = (int32_t)(jt * (WJ * RJ)) + w * RJ;
  const int32_t k0 = (int32_t)(kt * 128u) + lane * 2;
  const int32_t ib = (int32_t)ct * P.chunk;
  const int32_t ie = ib + P.chunk < P.N0 ? ib + P.chunk : P.N0;
  const int64_t plane = (int64_t)P.N1 * P.N2;
  const uint32_t plane_b = (uint32_t)(plane * 8);
  const uint32_t off = (uint32_t)(((int64_t)j0 * P.N2 + k0) * 8);  // out of range => the buffer returns 0 / drops the store
  const uint32_t rowb = (uint32_t)P.N2 * 8;
  auto rsrc = [&](const double* base, int32_t ip) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (int64_t)ip * plane), 0, plane_b, 0x00020000);
  };
  U4 nxt[RJ];
  {
    auto r = rsrc(P.in, ib);
    sfor<RJ>([&](auto rc) { constexpr int rr = rc; nxt[rr] = __builtin_amdgcn_raw_buffer_load_b128(r, off + rr * rowb, 0, LDAUX); });
  }
  for (int32_t i = ib; i < ie; ++i) {
    U4 cur[RJ];
    sfor<RJ>([&](auto rc) { constexpr int rr = rc; cur[rr] = nxt[rr]; });
    __syncthreads();
    if (i + 1 < ie) {
      auto r = rsrc(P.in, i + 1);
      sfor<RJ>([&](auto rc) { constexpr int rr = rc; nxt[rr] = __builtin_amdgcn_raw_buffer_load_b128(r, off + rr * rowb, 0, LDAUX); });
    }
    auto wr = rsrc(P.out, i);
    sfor<RJ>([&](auto rc) { constexpr int rr = rc; __builtin_amdgcn_raw_buffer_store_b128(cur[rr], wr, off + rr * rowb, 0, STAUX); });
  }
}
// the march pattern WITH the stencil's halo traffic, to price each piece: MODE bits
//   1: split row loads -- lanes 8..55 non-temporal, the two edge 128-byte lines (lanes 0..7, 56..63) with the default
//      policy (so that a K-neighbouring workgroup's halo request can find them in L2)
//   2: K-halo requests: lanes 0 and 63 load the 8 bytes just outside the wave's span, for plane i+1 (one step behind
//      the rows of that plane), default policy, vmcnt-tracked
//   4: J-halo rows: waves 0 and 15 load the row above / below the workgroup's tile (plane i+1)
//   8: all row loads non-temporal (without bit 1: all default)
template <int MODE>
__global__ __launch_bounds__(1024) void march_halo(P3 P, double* sink) {
  extern __shared__ char dyn_lds[];
  constexpr int RJ = 4, WJ = 16;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t v = xcd_remap(blockIdx.x, gridDim.x);
  const uint32_t kt = v % P.nK, t = v / P.nK, jt = t % P.nJ, ct = t / P.nJ;
  const int32_t j0 = (int32_t)(jt * (WJ * RJ)) + w * RJ;
  const int32_t kw = (int32_t)(kt * 128u), k0 = kw + lane * 2;
  const int32_t ib = (int32_t)ct * P.chunk;
  const int32_t ie = ib + P.chunk < P.N0 ? ib + P.chunk : P.N0;
  const int64_t plane = (int64_t)P.N1 * P.N2;
  const uint32_t plane_b = (uint32_t)(plane * 8);
  const uint32_t off = (uint32_t)(((int64_t)j0 * P.N2 + k0) * 8);
  const uint32_t rowb = (uint32_t)P.N2 * 8;
  // halo cell of this lane (only lanes 0 and 63 use it): k = kw-1 or kw+128, clamped into the row
  int32_t kh = lane == 0 ? kw - 1 : kw + 128;
  kh = kh < 0 ? 0 : (kh >= P.N2 ? P.N2 - 1 : kh);
  const uint32_t hoff = (uint32_t)(((int64_t)j0 * P.N2 + kh) * 8);
  const bool edge_lane = lane < 8 || lane >= 56, halo_lane = lane == 0 || lane == 63;
  int32_t jh = w == 0 ? j0 - 1 : j0 + RJ;
  jh = jh < 0 ? 0 : (jh >= P.N1 ? P.N1 - 1 : jh);
  const uint32_t jhoff = (uint32_t)(((int64_t)jh * P.N2 + k0) * 8);
  auto rsrc = [&](const double* base, int32_t ip) {
    ip = ip >= P.N0 ? P.N0 - 1 : ip;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (int64_t)ip * plane), 0, plane_b, 0x00020000);
  };
  typedef unsigned int U2 __attribute__((ext_vector_type(2)));
  U4 nxt[RJ], jrow = {0, 0, 0, 0};
  U2 kh_cur[RJ], kh_nxt[RJ];
  auto load_rows = [&](int32_t ip) {
    auto r = rsrc(P.in, ip);
    sfor<RJ>([&](auto rc) {
      constexpr int rr = rc;
      if constexpr (MODE & 1) {
        if (!edge_lane) nxt[rr] = __builtin_amdgcn_raw_buffer_load_b128(r, off + rr * rowb, 0, 2);
        if (edge_lane) nxt[rr] = __builtin_amdgcn_raw_buffer_load_b128(r, off + rr * rowb, 0, 0);
      } else {
        nxt[rr] = __builtin_amdgcn_raw_buffer_load_b128(r, off + rr * rowb, 0, (MODE & 8) ? 2 : 0);
      }
    });
  };
  auto load_halos = [&](int32_t ip) {
    auto r = rsrc(P.in, ip);
    if constexpr (MODE & 2) {
      sfor<RJ>([&](auto rc) {
        constexpr int rr = rc;
        if (halo_lane) kh_nxt[rr] = __builtin_amdgcn_raw_buffer_load_b64(r, hoff + rr * rowb, 0, 0);
      });
    }
    if constexpr (MODE & 4) {
      if (w == 0 || w == WJ - 1) jrow = __builtin_amdgcn_raw_buffer_load_b128(r, jhoff, 0, (MODE & 9) ? 2 : 0);
    }
  };
  sfor<RJ>([&](auto rc) { constexpr int rr = rc; kh_nxt[rr] = U2{0, 0}; kh_cur[rr] = U2{0, 0}; });
  load_rows(ib);
  load_halos(ib);
  U4 acc = {0, 0, 0, 0};
  for (int32_t i = ib; i < ie; ++i) {
    U4 cur[RJ];
    sfor<RJ>([&](auto rc) { constexpr int rr = rc; cur[rr] = nxt[rr]; kh_cur[rr] = kh_nxt[rr]; });
    acc += jrow;
    __syncthreads();
    if (i + 1 < ie) load_rows(i + 1);
    if (i + 1 < ie) load_halos(i + 1);
    auto wr = rsrc(P.out, i);
    sfor<RJ>([&](auto rc) {
      constexpr int rr = rc;
      if constexpr (MODE & 2) { acc.z += kh_cur[rr].x; acc.w += kh_cur[rr].y; }   // keeps the halo loads alive
      __builtin_amdgcn_raw_buffer_store_b128(cur[rr], wr, off + rr * rowb, 0, 2);
    });
  }
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u && acc.z == 77u && acc.w == 99u) *sink = 1.0;
}
template <int MODE>
static void run_halo(const char* what, const double* in, double* out, int N0, int N1, int N2, int chunk, int reps) {
  P3 P{in, out, N0, N1, N2, chunk, (uint32_t)((N1 + 63) / 64), (uint32_t)((N2 + 127) / 128), (uint32_t)((N0 + chunk - 1) / chunk), 0};
  const uint32_t blocks = P.nJ * P.nK * P.nC;
  const size_t lds = 159 * 1024;
  auto kern = march_halo<MODE>;
  CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  double* sink = out;  // never written (the condition cannot hold)
  const double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), lds, 0, P, sink); }, 2, reps);
  printf("march_halo mode=%2d %-58s chunk=%d %8.4f ms %7.1f GB/s\n", MODE, what, chunk, ms, 2.0 * N0 * (double)N1 * N2 * 8 / ms / 1e6);
  fflush(stdout);
}

template <int LDAUX, int STAUX>
static void run_buf(const double* in, double* out, int N0, int N1, int N2, int chunk, int reps) {
  P3 P{in, out, N0, N1, N2, chunk, (uint32_t)((N1 + 63) / 64), (uint32_t)((N2 + 127) / 128), (uint32_t)((N0 + chunk - 1) / chunk), 0};
  const uint32_t blocks = P.nJ * P.nK * P.nC;
  const size_t lds = 159 * 1024;
  auto kern = march_copy_buf<LDAUX, STAUX>;
  CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), lds, 0, P); }, 2, reps);
  auto nm = [](int a) { return a == 0 ? "plain" : a == 1 ? "sc0" : a == 2 ? "nt" : a == 3 ? "sc0 nt" : a == 16 ? "sc1" : a == 17 ? "sc0 sc1" : a == 18 ? "nt sc1" : "sc0 nt sc1"; };
  printf("march_copy_buf rj4_wj16_wk1_pf1 chunk=%d  load[%-10s] store[%-10s]  %8.4f ms %7.1f GB/s\n", chunk, nm(LDAUX), nm(STAUX), ms,
         2.0 * N0 * (double)N1 * N2 * 8 / ms / 1e6);
  fflush(stdout);
}

// ragged rows, store side: the result rows start 8 bytes off a 16-byte boundary.  MODE 0: unaligned 16-byte stores (what
// the march kernel did in round 1); MODE 1: results moved one cell across lanes (DPP) so that lanes 1..63 store ALIGNED
// 16-byte vectors, lane 0 and lane 63 store the two 8-byte end pieces of the wave's span
template <int MODE>
__global__ __launch_bounds__(1024) void march_copy_ragged(P3 P) {
  extern __shared__ char dyn_lds[];
  constexpr int RJ = 4, WJ = 16;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t v = xcd_remap(blockIdx.x, gridDim.x);
  const uint32_t kt = v % P.nK, t = v / P.nK, jt = t % P.nJ, ct = t / P.nJ;
  const int32_t j0 = (int32_t)(jt * (WJ * RJ)) + w * RJ;
  const int32_t k0 = (int32_t)(kt * 128u) + lane * 2;
  const int32_t ib = (int32_t)ct * P.chunk;
  const int32_t ie = ib + P.chunk < P.N0 ? ib + P.chunk : P.N0;
  const int64_t plane = (int64_t)P.N1 * P.N2;
  const int64_t base = (int64_t)j0 * P.N2 + k0;
  V2 nxt[RJ];
  sfor<RJ>([&](auto rc) { constexpr int r = rc; nxt[r] = *reinterpret_cast<const V2*>(P.in + (int64_t)ib * plane + base + (int64_t)r * P.N2); });
  for (int32_t i = ib; i < ie; ++i) {
    V2 cur[RJ];
    sfor<RJ>([&](auto rc) { constexpr int r = rc; cur[r] = nxt[r]; });
    __syncthreads();
    if (i + 1 < ie)
      sfor<RJ>([&](auto rc) { constexpr int r = rc; nxt[r] = *reinterpret_cast<const V2*>(P.in + (int64_t)(i + 1) * plane + base + (int64_t)r * P.N2); });
    sfor<RJ>([&](auto rc) {
      constexpr int r = rc;
      double* dst = P.out + (int64_t)i * plane + base + (int64_t)r * P.N2;   // 8 bytes off a 16-byte boundary
      if constexpr (MODE == 0) {
        typedef double UV2 __attribute__((ext_vector_type(2), aligned(8)));
        __builtin_nontemporal_store(cur[r], reinterpret_cast<V2*>(dst));
      } else {
        // the aligned vector below my cells holds (previous lane's second cell, my first cell)
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(cur[r].y), 0x138, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(cur[r].y), 0x138, 0xf, 0xf, false);
        V2 al = {__hiloint2double(hi, lo), cur[r].x};
        if (lane > 0) __builtin_nontemporal_store(al, reinterpret_cast<V2*>(dst - 1));
        else __builtin_nontemporal_store(cur[r].x, dst);
        if (lane == 63) __builtin_nontemporal_store(cur[r].y, dst + 1);
      }
    });
  }
}
template <int MODE>
static void run_ragged(const char* what, const double* in, double* out, int N0, int N1, int N2, int reps) {
  P3 P{in, out, N0, N1, N2, 128, (uint32_t)((N1 + 63) / 64), (uint32_t)((N2 + 127) / 128), (uint32_t)((N0 + 127) / 128), 0};
  const uint32_t blocks = P.nJ * P.nK * P.nC;
  const size_t lds = 159 * 1024;
  auto kern = march_copy_ragged<MODE>;
  CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), lds, 0, P); }, 2, reps);
  printf("march_copy_ragged %-60s %8.4f ms %7.1f GB/s\n", what, ms, 2.0 * N0 * (double)N1 * N2 * 8 / ms / 1e6);
  fflush(stdout);
}

// linear copies: U x 16 B per lane, exact grid
template <int U, bool NT>
__global__ __launch_bounds__(256) void lin_copy(const V2* __restrict__ src, V2* __restrict__ dst, int64_t n16) {
  const int64_t base = (int64_t)blockIdx.x * (256 * U) + threadIdx.x;
  V2 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = base + (int64_t)u * 256;
    if (i < n16) v[u] = src[i];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = base + (int64_t)u * 256;
    if (i < n16) {
      if (NT) __builtin_nontemporal_store(v[u], dst + i);
      else dst[i] = v[u];
    }
  }
}
// persistent linear copy: each workgroup streams one contiguous span, DEPTH x 4 KiB in flight per workgroup
template <int DEPTH, bool NT>
__global__ __launch_bounds__(256) void span_copy(const V2* __restrict__ src, V2* __restrict__ dst, int64_t n16) {
  const int64_t per = (n16 + gridDim.x - 1) / gridDim.x / 256 * 256 + 256;
  const int64_t b = (int64_t)blockIdx.x * per, e = b + per < n16 ? b + per : n16;
  for (int64_t i = b + threadIdx.x; i < e; i += 256 * DEPTH) {
    V2 v[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u)
      if (i + u * 256 < e) v[u] = src[i + u * 256];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u)
      if (i + u * 256 < e) {
        if (NT) __builtin_nontemporal_store(v[u], dst + i + u * 256);
        else dst[i + u * 256] = v[u];
      }
  }
}
// read-only and write-only streams
__global__ __launch_bounds__(256) void read_only(const V2* __restrict__ src, double* sink, int64_t n16) {
  const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  V2 a = {0, 0};
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t i = base + u * 256;
    if (i < n16) a += src[i];
  }
  if (a.x + a.y == 1.2345e300) *sink = a.x;
}
__global__ __launch_bounds__(256) void write_only(V2* __restrict__ dst, int64_t n16) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  V2 v = {1.0, 2.0};
  if (i < n16) __builtin_nontemporal_store(v, dst + i);
}

static hipEvent_t e0, e1;
template <class L>
static double time_ms(L&& launch, int warm, int reps) {
  for (int i = 0; i < warm; ++i) launch();
  CHECK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  CHECK(hipGetLastError());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

// wg_per_cu > 0: pad the launch with dynamic LDS so that at most that many workgroups fit a CU (160 KiB LDS)
template <int RJ, int WJ, int WK, int PF, bool SYNC, bool REMAP, bool NT, bool NTL = false>
static void run_march(const char* name, const double* in, double* out, int N0, int N1, int N2, int chunk, int reps,
                      int order = 0, int wg_per_cu = 0) {
  P3 P{in, out, N0, N1, N2, chunk, (uint32_t)((N1 + WJ * RJ - 1) / (WJ * RJ)), (uint32_t)((N2 + WK * 128 - 1) / (WK * 128)),
       (uint32_t)((N0 + chunk - 1) / chunk), order};
  const uint32_t blocks = P.nJ * P.nK * P.nC;
  const size_t lds = wg_per_cu > 0 ? (size_t)(160 * 1024 / wg_per_cu) - 1024 : 0;
  auto kern = march_copy<RJ, WJ, WK, PF, SYNC, REMAP, NT, NTL>;
  if (lds > 64 * 1024) CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = 0;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kWave * WJ * WK, lds));
  const double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(kWave * WJ * WK), lds, 0, P); }, 2, reps);
  const double gb = 2.0 * N0 * (double)N1 * N2 * 8 / ms / 1e6;
  printf("march_copy %-34s chunk=%4d order=%d wgs=%6u wg/cu=%d  %8.4f ms %7.1f GB/s\n", name, chunk, order, blocks, per_cu, ms, gb);
  fflush(stdout);
}

// timeline of `launches` back-to-back launches of one march_copy shape: when workgroups start, issue their first stores and end
// (100 MHz wall clock, all relative to the first start of each launch), and the gap to the next launch's first start
template <int RJ, int WJ, int WK, int PF>
static void run_timeline(const char* name, const double* in, double* out, int N0, int N1, int N2, int chunk, int launches, int order = 0) {
  P3 P{in, out, N0, N1, N2, chunk, (uint32_t)((N1 + WJ * RJ - 1) / (WJ * RJ)), (uint32_t)((N2 + WK * 128 - 1) / (WK * 128)),
       (uint32_t)((N0 + chunk - 1) / chunk), order, nullptr};
  const uint32_t blocks = P.nJ * P.nK * P.nC;
  uint64_t* ts;
  CHECK(hipMalloc(&ts, (size_t)launches * blocks * 3 * 8));
  auto kern = march_copy<RJ, WJ, WK, PF, true, true, true, false>;
  for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(kWave * WJ * WK), 0, 0, P);   // warm
  for (int l = 0; l < launches; ++l) {
    P.ts = ts + (size_t)l * blocks * 3;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(kWave * WJ * WK), 0, 0, P);
  }
  CHECK(hipDeviceSynchronize());
  std::vector<uint64_t> h((size_t)launches * blocks * 3);
  CHECK(hipMemcpy(h.data(), ts, h.size() * 8, hipMemcpyDeviceToHost));
  printf("timeline %s chunk=%d order=%d wgs=%u (us; 100 MHz clock)\n", name, chunk, order, blocks);
  uint64_t prev_end = 0;
  for (int l = 0; l < launches; ++l) {
    const uint64_t* t = h.data() + (size_t)l * blocks * 3;
    uint64_t s0 = ~0ull, s1 = 0, f0 = ~0ull, f1 = 0, e0 = ~0ull, e1 = 0;
    double esum = 0;
    for (uint32_t b = 0; b < blocks; ++b) {
      s0 = t[3 * b] < s0 ? t[3 * b] : s0; s1 = t[3 * b] > s1 ? t[3 * b] : s1;
      f0 = t[3 * b + 1] < f0 ? t[3 * b + 1] : f0; f1 = t[3 * b + 1] > f1 ? t[3 * b + 1] : f1;
      e0 = t[3 * b + 2] < e0 ? t[3 * b + 2] : e0; e1 = t[3 * b + 2] > e1 ? t[3 * b + 2] : e1;
    }
    for (uint32_t b = 0; b < blocks; ++b) esum += (double)(t[3 * b + 2] - s0);
    printf("  launch %2d: starts 0..%.2f  first step done %.2f..%.2f  ends %.2f..%.2f (mean %.2f)  span %.2f  gap since previous end %.2f\n", l,
           (s1 - s0) / 100.0, (f0 - s0) / 100.0, (f1 - s0) / 100.0, (e0 - s0) / 100.0, (e1 - s0) / 100.0, esum / blocks / 100.0,
           (e1 - s0) / 100.0, prev_end ? (double)((int64_t)(s0 - prev_end)) / 100.0 : 0.0);
    prev_end = e1;
    if (l == launches - 1) {
      // who finishes late?  mean end per XCD (blockIdx % 8) and per chunk index (virtual id / tiles)
      double xs[8] = {0}, xn[8] = {0};
      std::vector<double> cs(P.nC, 0.0), cn(P.nC, 0.0);
      for (uint32_t b = 0; b < blocks; ++b) {
        const double e = (double)(t[3 * b + 2] - s0) / 100.0;
        xs[b % 8] += e; xn[b % 8] += 1;
        const uint32_t v = xcd_remap_host(b, blocks), ct = P.order == 2 ? v % P.nC : v / (P.nJ * P.nK);
        cs[ct] += e; cn[ct] += 1;
      }
      printf("    mean end per XCD:  ");
      for (int x = 0; x < 8; ++x) printf(" %.1f", xs[x] / (xn[x] > 0 ? xn[x] : 1));
      printf("\n    mean end per chunk:");
      for (uint32_t c = 0; c < P.nC; ++c) printf(" %.1f", cs[c] / (cn[c] > 0 ? cn[c] : 1));
      printf("\n");
    }
  }
  CHECK(hipFree(ts));
}

// ---- pairs of workgroups meeting in the middle (what DESIGN.md section 8 item 3 proposes for the march kernel), tried on the copy ----
// Chunks 2m and 2m+1 of a tile form one plane range; the even workgroup marches UP from its bottom, the odd one DOWN from its top.
// mode 0: they meet at the static middle.  mode 1: when a workgroup has done 70 % of its static share it publishes how long that took;
// the second of the pair to get there knows both rates, splits the range in their proportion and publishes the split (one CAS); the
// first adopts it when it sees it (wave 0 polls one word per step, LDS carries it to the other waves behind the step's barrier) and
// falls back to the static middle if the word is still empty a few planes before it.  Nobody ever waits for anybody.
struct PairCtl { unsigned long long first_ticks; unsigned int n_up_plus1; unsigned int pad; };
template <int RJ, int WJ, int PF>
__global__ __launch_bounds__(kWave* WJ) void march_pair_copy(P3 P, PairCtl* ctl, int mode) {
  __shared__ int s_n;
  const int lane = threadIdx.x & 63;
  const int wj = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t v = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned long long t_start = wall_clock64();
  if (P.ts && threadIdx.x == 0) P.ts[3 * blockIdx.x] = t_start;
  const uint32_t kt = v % P.nK, tt = v / P.nK, jt = tt % P.nJ, ct = tt / P.nJ;
  const uint32_t seg = ct / 2, role = ct % 2;
  const int32_t sb = (int32_t)seg * 2 * P.chunk;
  const int32_t se = sb + 2 * P.chunk < P.N0 ? sb + 2 * P.chunk : P.N0;
  const int32_t S = se - sb, n_up_static = P.chunk < S ? P.chunk : S;
  const int32_t n_static = role == 0 ? n_up_static : S - n_up_static;
  PairCtl* c = ctl + ((size_t)seg * P.nJ * P.nK + (size_t)jt * P.nK + kt);
  const int32_t j0 = (int32_t)(jt * (WJ * RJ)) + wj * RJ;
  const int32_t k0 = (int32_t)(kt * 128u) + lane * 2;
  const bool ok = j0 < P.N1 && k0 < P.N2;
  const int64_t plane = (int64_t)P.N1 * P.N2;
  const int64_t base = ok ? (int64_t)j0 * P.N2 + k0 : 0;
  auto phys = [&](int32_t t) { return role == 0 ? sb + t : se - 1 - t; };
  if (threadIdx.x == 0) s_n = n_static;
  __syncthreads();
  int32_t n_cur = n_static;
  const int32_t t0 = (n_static * 7) / 10;
  bool decided = mode == 0 || S < 16;
  V2 nxt[PF][RJ];
  sfor<PF>([&](auto dc) {
    constexpr int d = dc;
    int32_t ip = phys(d < n_static ? d : (n_static > 0 ? n_static - 1 : 0));
    ip = ip < 0 ? 0 : (ip >= P.N0 ? P.N0 - 1 : ip);
    sfor<RJ>([&](auto rc) {
      constexpr int r = rc;
      nxt[d][r] = *reinterpret_cast<const V2*>(P.in + (int64_t)ip * plane + base + (int64_t)r * P.N2);
    });
  });
  auto step = [&](int32_t t, auto slot_c) {
    constexpr int slot = slot_c;
    V2 cur[RJ];
    sfor<RJ>([&](auto rc) { constexpr int r = rc; cur[r] = nxt[slot][r]; });
    if (threadIdx.x == 0 && !decided) {
      // wave 0, one lane: the pair's bookkeeping
      int32_t n_up = -1;
      if (t == t0) {
        unsigned long long el = wall_clock64() - t_start;
        el = el ? el : 1;
        const unsigned long long old = atomicCAS(&c->first_ticks, 0ull, el);
        if (old != 0) {   // second of the pair: split the range in the ratio of the two rates
          const double share_first = (double)el / (double)(old + el);       // the first one was faster: it gets the larger share
          int32_t n_first = (int32_t)(share_first * S + 0.5);
          const int32_t lo = t0 + PF + 3, hi = S - (t0 + PF + 3);
          n_first = n_first < lo ? lo : (n_first > hi ? hi : n_first);
          const int32_t want_up = role == 0 ? S - n_first : n_first;          // I am the second; the first has the other role
          const unsigned int prev = atomicCAS(&c->n_up_plus1, 0u, (unsigned int)want_up + 1u);
          n_up = prev ? (int32_t)prev - 1 : want_up;
        }
      } else if (t > t0) {
        const unsigned int seen = __hip_atomic_load(&c->n_up_plus1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen) n_up = (int32_t)seen - 1;
        else if (t + PF + 3 >= n_static) {   // nothing yet and the static middle is near: settle on it
          const unsigned int prev = atomicCAS(&c->n_up_plus1, 0u, (unsigned int)n_up_static + 1u);
          n_up = prev ? (int32_t)prev - 1 : n_up_static;
        }
      }
      if (n_up >= 0) { s_n = role == 0 ? n_up : S - n_up; decided = true; }
    }
    __syncthreads();
    n_cur = s_n;
    if (t + PF < n_cur) {
      int32_t ip = phys(t + PF);
      sfor<RJ>([&](auto rc) {
        constexpr int r = rc;
        nxt[slot][r] = *reinterpret_cast<const V2*>(P.in + (int64_t)ip * plane + base + (int64_t)r * P.N2);
      });
    }
    if (ok) {
      const int32_t op = phys(t);
      sfor<RJ>([&](auto rc) {
        constexpr int r = rc;
        __builtin_nontemporal_store(cur[r], reinterpret_cast<V2*>(P.out + (int64_t)op * plane + base + (int64_t)r * P.N2));
      });
    }
  };
  for (int32_t t = 0; t < n_cur; t += PF) {
    sfor<PF>([&](auto phc) {
      constexpr int ph = phc;
      if (t + ph < n_cur) step(t + ph, phc);
    });
  }
  if (P.ts && threadIdx.x == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    P.ts[3 * blockIdx.x + 1] = (unsigned long long)n_cur;
    P.ts[3 * blockIdx.x + 2] = wall_clock64();
  }
}

template <int RJ, int WJ, int PF>
static void run_pairs(const char* name, const double* in, double* out, int N0, int N1, int N2, int chunk, int mode, int reps) {
  P3 P{in, out, N0, N1, N2, chunk, (uint32_t)((N1 + WJ * RJ - 1) / (WJ * RJ)), (uint32_t)((N2 + 127) / 128), (uint32_t)((N0 + chunk - 1) / chunk), 0, nullptr};
  if (P.nC % 2) { printf("pairs: odd number of chunks\n"); return; }
  const uint32_t blocks = P.nJ * P.nK * P.nC;
  const size_t nctl = (size_t)(P.nC / 2) * P.nJ * P.nK;
  PairCtl* ctl;
  CHECK(hipMalloc(&ctl, nctl * sizeof(PairCtl)));
  auto kern = march_pair_copy<RJ, WJ, PF>;
  auto launch = [&] {
    (void)hipMemsetAsync(ctl, 0, nctl * sizeof(PairCtl), 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(kWave * WJ), 0, 0, P, ctl, mode);
  };
  const double ms = time_ms(launch, 5, reps);
  // one more launch with time stamps: spread of the ends, planes done by the up / down halves, and a check of the copy
  uint64_t* ts;
  CHECK(hipMalloc(&ts, (size_t)blocks * 3 * 8));
  CHECK(hipMemset(out, 0, (size_t)N0 * N1 * N2 * 8));
  P.ts = ts;
  launch();
  CHECK(hipDeviceSynchronize());
  std::vector<uint64_t> h((size_t)blocks * 3);
  CHECK(hipMemcpy(h.data(), ts, h.size() * 8, hipMemcpyDeviceToHost));
  uint64_t s0 = ~0ull, e0 = ~0ull, e1 = 0;
  double esum = 0;
  int nmin = 1 << 30, nmax = 0;
  for (uint32_t b = 0; b < blocks; ++b) {
    s0 = h[3 * b] < s0 ? h[3 * b] : s0;
    nmin = (int)h[3 * b + 1] < nmin ? (int)h[3 * b + 1] : nmin;
    nmax = (int)h[3 * b + 1] > nmax ? (int)h[3 * b + 1] : nmax;
  }
  for (uint32_t b = 0; b < blocks; ++b) {
    e0 = h[3 * b + 2] < e0 ? h[3 * b + 2] : e0; e1 = h[3 * b + 2] > e1 ? h[3 * b + 2] : e1;
    esum += (double)(h[3 * b + 2] - s0);
  }
  // the copy must be complete and exact whatever the splits were
  const size_t n = (size_t)N0 * N1 * N2;
  std::vector<double> hi(n), ho(n);
  CHECK(hipMemcpy(hi.data(), in, n * 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(ho.data(), out, n * 8, hipMemcpyDeviceToHost));
  const bool same = memcmp(hi.data(), ho.data(), n * 8) == 0;
  printf("pair_copy %-18s mode=%d (%s) chunk=%d wgs=%u  %8.4f ms %7.1f GB/s   ends %.1f..%.1f us (mean %.1f)  planes per workgroup %d..%d  copy %s\n", name, mode,
         mode ? "split by measured rates" : "static middle", chunk, blocks, ms, 2.0 * n * 8 / ms / 1e6, (e0 - s0) / 100.0, (e1 - s0) / 100.0,
         esum / blocks / 100.0, nmin, nmax, same ? "exact" : "WRONG");
  fflush(stdout);
  CHECK(hipFree(ts));
  CHECK(hipFree(ctl));
}

// short-lived workgroups (256 lanes) in linear order: a workgroup copies `nseg` segments of `seg16` 16-byte words at
// stride `stride16`; G = stride16 / seg16 consecutive workgroups interleave inside a super-block of nseg * stride16
// words, so the union over the grid is the whole buffer exactly once
__global__ __launch_bounds__(256) void seg_copy(const V2* __restrict__ src, V2* __restrict__ dst, int64_t n16, int seg16,
                                                int nseg, int64_t stride16) {
  const int64_t G = stride16 / seg16;
  const int64_t sb = blockIdx.x / G, g = blockIdx.x % G;
  const int64_t base = sb * nseg * stride16 + g * seg16;
  for (int s = 0; s < nseg; ++s)
    for (int64_t i = threadIdx.x; i < seg16; i += 256) {
      const int64_t a = base + s * stride16 + i;
      if (a < n16) __builtin_nontemporal_store(src[a], dst + a);
    }
}

__global__ __launch_bounds__(256) void fill_index(double* __restrict__ dst, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = (double)i * 0.5 + 1.0;
}

int main(int argc, char** argv) {
  int N0 = 1024, N1 = 1024, N2 = 1024;
  if (argc >= 4) { N0 = atoi(argv[1]); N1 = atoi(argv[2]); N2 = atoi(argv[3]); }
  const int reps = argc >= 5 ? atoi(argv[4]) : 10;
  const int64_t n = (int64_t)N0 * N1 * N2, n16 = n / 2;
  double *a, *b;
  CHECK(hipMalloc(&a, n * 8));
  CHECK(hipMalloc(&b, n * 8));
  CHECK(hipMemset(a, 0x3c, n * 8));
  CHECK(hipMemset(b, 0, n * 8));
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const double bytes2 = 2.0 * n * 8;
  if (argc >= 6 && !strcmp(argv[5], "timeline")) {
    time_ms([&] { hipLaunchKernelGGL((lin_copy<1, true>), dim3((n16 + 255) / 256), dim3(256), 0, 0, (const V2*)a, (V2*)b, n16); }, 0, 200);
    run_timeline<4, 8, 1, 1>("rj4_wj8_wk1_pf1", a, b, N0, N1, N2, N0 / 8 > 0 ? N0 / 8 : 1, 6);
    run_timeline<4, 8, 1, 2>("rj4_wj8_wk1_pf2", a, b, N0, N1, N2, N0 / 8 > 0 ? N0 / 8 : 1, 6);
    run_timeline<4, 4, 2, 2>("rj4_wj4_wk2_pf2", a, b, N0, N1, N2, N0 / 8 > 0 ? N0 / 8 : 1, 6);
    run_timeline<4, 8, 1, 1>("rj4_wj8_wk1_pf1", a, b, N0, N1, N2, N0 / 16 > 0 ? N0 / 16 : 1, 6);
    run_timeline<4, 8, 1, 1>("rj4_wj8_wk1_pf1 chunks fastest (every XCD works on every chunk)", a, b, N0, N1, N2, N0 / 8 > 0 ? N0 / 8 : 1, 6, 2);
    return 0;
  }
  if (argc >= 6 && !strcmp(argv[5], "pairs")) {
    time_ms([&] { hipLaunchKernelGGL((lin_copy<1, true>), dim3((n16 + 255) / 256), dim3(256), 0, 0, (const V2*)a, (V2*)b, n16); }, 0, 200);
    hipLaunchKernelGGL(fill_index, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, a, n);   // every cell different: a misplaced plane shows
    for (int rep = 0; rep < 3; ++rep) {
      run_march<4, 8, 1, 1, true, true, true>("rj4_wj8_wk1_pf1 sync (one workgroup per chunk, today)", a, b, N0, N1, N2, N0 / 8, reps, 0, 1);
      run_pairs<4, 8, 1>("rj4_wj8_pf1", a, b, N0, N1, N2, N0 / 8, 0, reps);
      run_pairs<4, 8, 1>("rj4_wj8_pf1", a, b, N0, N1, N2, N0 / 8, 1, reps);
      run_pairs<4, 8, 2>("rj4_wj8_pf2", a, b, N0, N1, N2, N0 / 8, 0, reps);
      run_pairs<4, 8, 2>("rj4_wj8_pf2", a, b, N0, N1, N2, N0 / 8, 1, reps);
    }
    return 0;
  }
  // ramp the clocks
  time_ms([&] { hipLaunchKernelGGL((lin_copy<1, true>), dim3((n16 + 255) / 256), dim3(256), 0, 0, (const V2*)a, (V2*)b, n16); }, 0, 40);
  for (int pass = 0; pass < 2; ++pass) {
    double ms;
    ms = time_ms([&] { hipLaunchKernelGGL((lin_copy<1, true>), dim3((n16 + 255) / 256), dim3(256), 0, 0, (const V2*)a, (V2*)b, n16); }, 2, reps);
    printf("lin_copy U1 nt                      %8.4f ms %7.1f GB/s\n", ms, bytes2 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((lin_copy<1, false>), dim3((n16 + 255) / 256), dim3(256), 0, 0, (const V2*)a, (V2*)b, n16); }, 2, reps);
    printf("lin_copy U1 plain                   %8.4f ms %7.1f GB/s\n", ms, bytes2 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((lin_copy<4, true>), dim3((n16 + 1023) / 1024), dim3(256), 0, 0, (const V2*)a, (V2*)b, n16); }, 2, reps);
    printf("lin_copy U4 nt                      %8.4f ms %7.1f GB/s\n", ms, bytes2 / ms / 1e6);
    for (int g : {256, 512, 1024, 2048, 4096}) {
      ms = time_ms([&] { hipLaunchKernelGGL((span_copy<4, true>), dim3(g), dim3(256), 0, 0, (const V2*)a, (V2*)b, n16); }, 2, reps);
      printf("span_copy D4 nt grid=%5d           %8.4f ms %7.1f GB/s\n", g, ms, bytes2 / ms / 1e6);
    }
    ms = time_ms([&] { hipLaunchKernelGGL((span_copy<8, true>), dim3(2048), dim3(256), 0, 0, (const V2*)a, (V2*)b, n16); }, 2, reps);
    printf("span_copy D8 nt grid= 2048           %8.4f ms %7.1f GB/s\n", ms, bytes2 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(read_only, dim3((n16 + 1023) / 1024), dim3(256), 0, 0, (const V2*)a, b, n16); }, 2, reps);
    printf("read_only                           %8.4f ms %7.1f GB/s (one field)\n", ms, bytes2 / 2 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(write_only, dim3((n16 + 255) / 256), dim3(256), 0, 0, (V2*)b, n16); }, 2, reps);
    printf("write_only nt                       %8.4f ms %7.1f GB/s (one field)\n", ms, bytes2 / 2 / ms / 1e6);
    fflush(stdout);
  }
  {
    // short-lived workgroups: per-workgroup access shape in isolation
    struct { int seg16, nseg; int64_t stride16; const char* what; } shapes[] = {
        {256, 1, 256, "1 x 4 KiB (= lin_copy U1)"},      {64, 4, 512, "4 x 1 KiB @ 8 KiB"},
        {64, 16, 512, "16 x 1 KiB @ 8 KiB"},             {64, 64, 512, "64 x 1 KiB @ 8 KiB (a wj16 tile's plane)"},
        {128, 32, 512, "32 x 2 KiB @ 8 KiB"},            {512, 8, 512, "8 x 8 KiB contiguous (64 KiB)"},
        {256, 16, 512, "16 x 4 KiB @ 8 KiB"},            {64, 64, 64 * 8192, "64 x 1 KiB @ 8 MiB (a wave column over 64 planes)"},
    };
    for (auto& sh : shapes) {
      const int64_t per_wg = (int64_t)sh.seg16 * sh.nseg;
      const double ms = time_ms([&] { hipLaunchKernelGGL(seg_copy, dim3((uint32_t)((n16 + per_wg - 1) / per_wg)), dim3(256), 0, 0, (const V2*)a, (V2*)b, n16, sh.seg16, sh.nseg, sh.stride16); }, 2, reps);
      printf("seg_copy %-52s %8.4f ms %7.1f GB/s\n", sh.what, ms, bytes2 / ms / 1e6);
      fflush(stdout);
    }
  }
  // the march pattern at the stencil's residency (1 workgroup of 16 waves per CU, or 2 of 8, 4 of 4): chunk lengths
  for (int chunk : {16, 32, 64, 128, 256, 1024}) {
    run_march<4, 16, 1, 1, true, true, true>("rj4_wj16_wk1_pf1 sync", a, b, N0, N1, N2, chunk, reps, 0, 1);
    run_march<4, 8, 1, 1, true, true, true>("rj4_wj8_wk1_pf1 sync", a, b, N0, N1, N2, chunk, reps, 0, 2);
    run_march<4, 4, 1, 1, true, true, true>("rj4_wj4_wk1_pf1 sync", a, b, N0, N1, N2, chunk, reps, 0, 4);
  }
  // tile numbering
  for (int order : {0, 1, 2}) {
    run_march<4, 16, 1, 1, true, true, true>("rj4_wj16_wk1_pf1 sync", a, b, N0, N1, N2, 128, reps, order, 1);
    run_march<4, 16, 1, 1, true, false, true>("rj4_wj16_wk1_pf1 sync noremap", a, b, N0, N1, N2, 128, reps, order, 1);
  }
  // residency and the other knobs
  run_march<4, 16, 1, 1, true, true, true>("rj4_wj16_wk1_pf1 sync", a, b, N0, N1, N2, 128, reps, 0, 2);
  run_march<4, 16, 1, 2, true, true, true>("rj4_wj16_wk1_pf2 sync", a, b, N0, N1, N2, 128, reps, 0, 1);
  run_march<4, 16, 1, 1, true, true, true, true>("rj4_wj16_wk1_pf1 sync ntload", a, b, N0, N1, N2, 128, reps, 0, 1);
  run_march<4, 16, 1, 1, true, true, false>("rj4_wj16_wk1_pf1 sync plainstore", a, b, N0, N1, N2, 128, reps, 0, 1);
  run_march<4, 8, 2, 1, true, true, true>("rj4_wj8_wk2_pf1 sync", a, b, N0, N1, N2, 128, reps, 0, 1);
  run_march<4, 2, 8, 1, true, true, true>("rj4_wj2_wk8_pf1 sync", a, b, N0, N1, N2, 128, reps, 0, 1);
  run_march<8, 8, 1, 1, true, true, true>("rj8_wj8_wk1_pf1 sync", a, b, N0, N1, N2, 128, reps, 0, 1);
  run_march<2, 16, 1, 1, true, true, true>("rj2_wj16_wk1_pf1 sync", a, b, N0, N1, N2, 128, reps, 0, 1);
  run_march<2, 16, 1, 2, true, true, true>("rj2_wj16_wk1_pf2 sync", a, b, N0, N1, N2, 128, reps, 0, 1);
  run_march<1, 16, 1, 4, true, true, true>("rj1_wj16_wk1_pf4 sync", a, b, N0, N1, N2, 128, reps, 0, 1);
  run_march<4, 16, 1, 1, false, true, true>("rj4_wj16_wk1_pf1 nosync", a, b, N0, N1, N2, 128, reps, 0, 1);
  // cache-policy flavours of the loads and stores (buffer instructions)
  run_buf<0, 2>(a, b, N0, N1, N2, 128, reps);
  run_buf<2, 2>(a, b, N0, N1, N2, 128, reps);
  run_buf<16, 2>(a, b, N0, N1, N2, 128, reps);
  run_buf<17, 2>(a, b, N0, N1, N2, 128, reps);
  run_buf<18, 2>(a, b, N0, N1, N2, 128, reps);
  run_buf<19, 2>(a, b, N0, N1, N2, 128, reps);
  run_buf<1, 2>(a, b, N0, N1, N2, 128, reps);
  run_buf<2, 0>(a, b, N0, N1, N2, 128, reps);
  run_buf<2, 16>(a, b, N0, N1, N2, 128, reps);
  run_buf<2, 17>(a, b, N0, N1, N2, 128, reps);
  run_buf<2, 18>(a, b, N0, N1, N2, 128, reps);
  run_buf<2, 19>(a, b, N0, N1, N2, 128, reps);
  run_buf<2, 3>(a, b, N0, N1, N2, 128, reps);
  run_buf<0, 2>(a, b, N0, N1, N2, 128, reps);
  run_buf<2, 2>(a, b, N0, N1, N2, 128, reps);
  // what each piece of the stencil's halo traffic costs on top of the bare pattern
  for (int rep = 0; rep < 2; ++rep) {
    run_halo<0>("rows default policy", a, b, N0, N1, N2, 128, reps);
    run_halo<8>("rows nt", a, b, N0, N1, N2, 128, reps);
    run_halo<1>("rows split: edge lines default, interior nt", a, b, N0, N1, N2, 128, reps);
    run_halo<2>("rows default + K-halo requests", a, b, N0, N1, N2, 128, reps);
    run_halo<3>("rows split + K-halo requests", a, b, N0, N1, N2, 128, reps);
    run_halo<10>("rows nt + K-halo requests", a, b, N0, N1, N2, 128, reps);
    run_halo<6>("rows default + K-halo + J-halo rows", a, b, N0, N1, N2, 128, reps);
    run_halo<7>("rows split + K-halo + J-halo rows (nt)", a, b, N0, N1, N2, 128, reps);
    run_halo<14>("rows nt + K-halo + J-halo rows (nt)", a, b, N0, N1, N2, 128, reps);
    run_halo<12>("rows nt + J-halo rows (nt)", a, b, N0, N1, N2, 128, reps);
  }
  // ragged rows: what 8-byte-misaligned 16-byte accesses cost on the load side and on the store side
  // (same traversal; the field pointers are moved by one double, the last plane is left out to stay in bounds)
  if (N0 > 2) {
    run_march<4, 16, 1, 1, true, true, true>("aligned loads, aligned stores", a, b, N0 - 1, N1, N2, 128, reps, 0, 1);
    run_march<4, 16, 1, 1, true, true, true>("MISALIGNED loads, aligned stores", a + 1, b, N0 - 1, N1, N2, 128, reps, 0, 1);
    run_march<4, 16, 1, 1, true, true, true>("aligned loads, MISALIGNED stores", a, b + 1, N0 - 1, N1, N2, 128, reps, 0, 1);
    run_march<4, 16, 1, 1, true, true, true>("MISALIGNED loads and stores", a + 1, b + 1, N0 - 1, N1, N2, 128, reps, 0, 1);
    run_march<4, 16, 1, 1, true, true, false>("aligned loads, MISALIGNED plain stores", a, b + 1, N0 - 1, N1, N2, 128, reps, 0, 1);
    run_ragged<0>("rows 8 B off: unaligned 16-byte stores", a, b + 1, N0 - 1, N1, N2, reps);
    run_ragged<1>("rows 8 B off: results shifted one cell, aligned stores + 8-byte ends", a, b + 1, N0 - 1, N1, N2, reps);
    run_ragged<0>("rows aligned: plain 16-byte stores (reference)", a, b, N0 - 1, N1, N2, reps);
    // which granule has to be written whole?  stores whose wave spans start 16 / 32 / 64 bytes off a 128-byte line
    run_march<4, 16, 1, 1, true, true, true>("stores +16 B", a, b + 2, N0 - 1, N1, N2, 128, reps, 0, 1);
    run_march<4, 16, 1, 1, true, true, true>("stores +32 B", a, b + 4, N0 - 1, N1, N2, 128, reps, 0, 1);
    run_march<4, 16, 1, 1, true, true, true>("stores +64 B", a, b + 8, N0 - 1, N1, N2, 128, reps, 0, 1);
    run_march<4, 16, 1, 1, true, true, true>("stores +0 B", a, b, N0 - 1, N1, N2, 128, reps, 0, 1);
    // do the halves of a sector that two waves of ONE workgroup write at the same time merge in L2?
    run_march<4, 8, 2, 1, true, true, true>("wk2 aligned", a, b, N0 - 1, N1, N2, 128, reps, 0, 1);
    run_march<4, 8, 2, 1, true, true, true>("wk2 MISALIGNED stores nt", a, b + 1, N0 - 1, N1, N2, 128, reps, 0, 1);
    run_march<4, 8, 2, 1, true, true, false>("wk2 MISALIGNED stores plain", a, b + 1, N0 - 1, N1, N2, 128, reps, 0, 1);
    run_march<4, 2, 8, 1, true, true, true>("wk8 aligned", a, b, N0 - 1, N1, N2, 128, reps, 0, 1);
    run_march<4, 2, 8, 1, true, true, true>("wk8 MISALIGNED stores nt", a, b + 1, N0 - 1, N1, N2, 128, reps, 0, 1);
    run_march<4, 2, 8, 1, true, true, false>("wk8 MISALIGNED stores plain", a, b + 1, N0 - 1, N1, N2, 128, reps, 0, 1);
    run_march<4, 4, 4, 1, true, true, true>("wk4 MISALIGNED stores nt", a, b + 1, N0 - 1, N1, N2, 128, reps, 0, 1);
    run_march<4, 4, 4, 1, true, true, false>("wk4 MISALIGNED stores plain", a, b + 1, N0 - 1, N1, N2, 128, reps, 0, 1);
  }
  CHECK(hipFree(a));
  CHECK(hipFree(b));
  return 0;
}

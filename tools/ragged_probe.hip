// ragged_probe.hip -- ragged rows (row bytes not a multiple of 64): does a copy with the march kernel's traversal reach
// the aligned rate when every wave stores WHOLE 64-byte granules?  Windows overlap along K (stride 120 of 128 fp64 cells)
// and each row keeps the cells between the first granule boundary at or after the window's nominal start and the first
// one at or after the next window's start; the two lanes a boundary cuts through store one 8-byte half.  Measurement tool
// only: nothing here is part of the product.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ragged_probe.hip -o build/ragged_probe && build/ragged_probe [N0 N1 N2]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <utility>
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
typedef double UV2 __attribute__((ext_vector_type(2), aligned(8)));

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
};

// MODE 0: windows side by side (stride 128), every lane stores its 16 bytes wherever they fall (today's ragged path)
// MODE 2: windows overlap (stride 120), whole granules per wave and row
// MODE 3: like 2 but plain (write-back) stores
template <int MODE, int RJ, int WJ>
__global__ __launch_bounds__(64 * WJ) void ragged_copy(P3 P) {
  constexpr int STRIDE = MODE == 0 ? 128 : 120;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t v = xcd_remap(blockIdx.x, gridDim.x);
  const uint32_t kt = v % P.nK, t = v / P.nK, jt = t % P.nJ, ct = t / P.nJ;
  const int32_t j0 = (int32_t)(jt * (WJ * RJ)) + w * RJ;
  const int32_t kw = (int32_t)(kt * STRIDE);
  const int32_t k0 = kw + lane * 2;
  // the last vector that may be loaded starts at N2-2; a lane at N2-1 (odd N2) takes its cell from that vector's .y
  const int32_t kl = k0 < P.N2 - 2 ? k0 : P.N2 - 2;
  const bool odd_tail = k0 == P.N2 - 1;
  const int32_t ib = (int32_t)ct * P.chunk;
  const int32_t ie = ib + P.chunk < P.N0 ? ib + P.chunk : P.N0;
  const int64_t plane = (int64_t)P.N1 * P.N2;
  int64_t rowc[RJ];   // first cell of own row r in plane 0 (clamped rows repeat the last one; their stores are off)
  bool row_ok[RJ];
  sfor<RJ>([&](auto rc) {
    constexpr int r = rc;
    const int32_t j = j0 + r < P.N1 ? j0 + r : P.N1 - 1;
    row_ok[r] = j0 + r < P.N1;
    rowc[r] = (int64_t)j * P.N2;
  });
  auto load = [&](int32_t i, int r) -> V2 {
    V2 x = *reinterpret_cast<const UV2*>(P.in + (int64_t)i * plane + rowc[r] + kl);
    if (odd_tail) x.x = x.y;
    return x;
  };
  V2 nxt[RJ];
  sfor<RJ>([&](auto rc) { constexpr int r = rc; nxt[r] = load(ib, r); });
  for (int32_t i = ib; i < ie; ++i) {
    V2 cur[RJ];
    sfor<RJ>([&](auto rc) { constexpr int r = rc; cur[r] = nxt[r]; });
    __syncthreads();
    if (i + 1 < ie) sfor<RJ>([&](auto rc) { constexpr int r = rc; nxt[r] = load(i + 1, r); });
    sfor<RJ>([&](auto rc) {
      constexpr int r = rc;
      const int64_t rs = (int64_t)i * plane + rowc[r];
      double* dst = P.out + rs + k0;
      if constexpr (MODE == 0) {
        if (row_ok[r]) {
          if (k0 + 1 < P.N2) __builtin_nontemporal_store(cur[r], reinterpret_cast<UV2*>(dst));
          else if (k0 < P.N2) __builtin_nontemporal_store(cur[r].x, dst);
        }
      } else {
        // kept cells [c0, c1): granule boundaries (8 cells) of THIS row at or after the nominal window starts
        const int32_t a = (int32_t)(rs & 7);                 // cells the row start lies past a granule boundary
        int32_t c0 = kw == 0 ? 0 : kw + ((8 - ((a + kw) & 7)) & 7);
        int32_t c1 = kw + STRIDE + ((8 - ((a + kw + STRIDE) & 7)) & 7);
        c0 = c0 < P.N2 ? c0 : P.N2;
        c1 = c1 < P.N2 ? c1 : P.N2;
        const bool kx = k0 >= c0 && k0 < c1, ky = k0 + 1 >= c0 && k0 + 1 < c1;
        if (row_ok[r]) {
          if constexpr (MODE == 2) {
            if (kx && ky) __builtin_nontemporal_store(cur[r], reinterpret_cast<UV2*>(dst));
            else if (kx) __builtin_nontemporal_store(cur[r].x, dst);
            else if (ky) __builtin_nontemporal_store(cur[r].y, dst + 1);
          } else {
            if (kx && ky) *reinterpret_cast<UV2*>(dst) = cur[r];
            else if (kx) dst[0] = cur[r].x;
            else if (ky) dst[1] = cur[r].y;
          }
        }
      }
    });
  }
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

__global__ void fill(double* p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = (double)(i % 1000003);
}
__global__ void count_diff(const double* a, const double* b, int64_t n, unsigned long long* cnt) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n && a[i] != b[i]) atomicAdd(cnt, 1ull);
}

template <int MODE, int RJ, int WJ>
static void run(const char* what, const double* in, double* out, int N0, int N1, int N2, int chunk, int reps,
                unsigned long long* cnt) {
  constexpr int STRIDE = MODE == 0 ? 128 : 120;
  P3 P{in, out, N0, N1, N2, chunk, (uint32_t)((N1 + WJ * RJ - 1) / (WJ * RJ)), (uint32_t)((N2 + STRIDE - 1) / STRIDE),
       (uint32_t)((N0 + chunk - 1) / chunk)};
  const uint32_t blocks = P.nJ * P.nK * P.nC;
  const int64_t n = (int64_t)N0 * N1 * N2;
  CHECK(hipMemset(out, 0xff, n * 8));
  auto kern = ragged_copy<MODE, RJ, WJ>;
  const double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * WJ), 0, 0, P); }, 2, reps);
  CHECK(hipMemset(cnt, 0, 8));
  hipLaunchKernelGGL(count_diff, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, in, (const double*)out, n, cnt);
  unsigned long long bad = 0;
  CHECK(hipMemcpy(&bad, cnt, 8, hipMemcpyDeviceToHost));
  printf("ragged_copy %-58s rj%d wj%-2d chunk=%4d wgs=%6u  %8.4f ms %7.1f GB/s  mismatches=%llu\n", what, RJ, WJ, chunk, blocks, ms,
         2.0 * n * 8 / ms / 1e6, bad);
  fflush(stdout);
}

int main(int argc, char** argv) {
  int N0 = 1025, N1 = 1025, N2 = 1025;
  if (argc >= 4) { N0 = atoi(argv[1]); N1 = atoi(argv[2]); N2 = atoi(argv[3]); }
  const int reps = argc >= 5 ? atoi(argv[4]) : 10;
  const int64_t n = (int64_t)N0 * N1 * N2;
  double *a, *b;
  unsigned long long* cnt;
  CHECK(hipMalloc(&a, n * 8 + 64));
  CHECK(hipMalloc(&b, n * 8 + 64));
  CHECK(hipMalloc(&cnt, 8));
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(fill, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, a, n);
  CHECK(hipDeviceSynchronize());
  // ramp the clocks
  run<0, 4, 16>("ramp", a, b, N0, N1, N2, 128, 40, cnt);
  for (int pass = 0; pass < 2; ++pass) {
    for (int chunk : {128, 256}) {
      run<0, 4, 16>("side by side, unaligned 16-byte nt stores (today)", a, b, N0, N1, N2, chunk, reps, cnt);
      run<2, 4, 16>("overlapping windows, whole granules, nt stores", a, b, N0, N1, N2, chunk, reps, cnt);
      run<3, 4, 16>("overlapping windows, whole granules, plain stores", a, b, N0, N1, N2, chunk, reps, cnt);
      run<0, 8, 8>("side by side, unaligned 16-byte nt stores (today)", a, b, N0, N1, N2, chunk, reps, cnt);
      run<2, 8, 8>("overlapping windows, whole granules, nt stores", a, b, N0, N1, N2, chunk, reps, cnt);
    }
  }
  CHECK(hipFree(a));
  CHECK(hipFree(b));
  return 0;
}

// ipc_probe.hip -- what the CU-free halo transport can be built from on this pool: two processes (forked BEFORE any HIP
// call) on one device, device memory shared through hipIpc{Get,Open}MemHandle, copies into the peer's mapping, flag kernels
// against a mailbox in the peer's memory, hipStream{Write,Wait}Value64, interprocess events.  Every device-side wait is
// bounded by the constant 100 MHz wall clock.  Measurement tool only: nothing here is part of the product.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ipc_probe.hip -o build/ipc_probe && build/ipc_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#define CHECK(e)                                                                                     \
  do {                                                                                               \
    hipError_t _e = (e);                                                                             \
    if (_e != hipSuccess) {                                                                          \
      fprintf(stderr, "[p%d] %s failed: %s (%s:%d)\n", g_me, #e, hipGetErrorString(_e), __FILE__, __LINE__); \
      exit(1);                                                                                       \
    }                                                                                                \
  } while (0)
#define SOFT(e)                                                                                      \
  ([&] {                                                                                             \
    hipError_t _e = (e);                                                                             \
    if (_e != hipSuccess) {                                                                          \
      fprintf(stderr, "[p%d] %s -> %s\n", g_me, #e, hipGetErrorString(_e));                          \
      (void)hipGetLastError();                                                                       \
    }                                                                                                \
    return _e == hipSuccess;                                                                         \
  }())

static int g_me = 0;
static int g_sock = -1;

static void xsend(const void* p, size_t n) {
  if (write(g_sock, p, n) != (ssize_t)n) { perror("write"); exit(1); }
}
static void xrecv(void* p, size_t n) {
  size_t got = 0;
  while (got < n) {
    ssize_t r = read(g_sock, (char*)p + got, n - got);
    if (r <= 0) { fprintf(stderr, "[p%d] peer closed\n", g_me); exit(1); }
    got += (size_t)r;
  }
}
static void host_barrier() { char c = 1; xsend(&c, 1); xrecv(&c, 1); }
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

__global__ void k_fill(uint64_t* p, size_t n, uint64_t tag) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = tag + i;
}
__global__ void k_signal(uint64_t* flag, uint64_t v) { __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void k_wait_ge(uint64_t* flag, uint64_t want, uint64_t timeout_ticks, uint32_t* err) {
  const uint64_t t0 = wall_clock64();
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
    if (wall_clock64() - t0 > timeout_ticks) { *err = 1; return; }
    __builtin_amdgcn_s_sleep(8);
  }
}
__global__ void k_check(const uint64_t* p, size_t n, uint64_t tag, uint32_t* bad) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n && p[i] != tag + i) atomicAdd(bad, 1u);
}

int main(int argc, char** argv) {
  int sv[2];
  if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv) != 0) { perror("socketpair"); return 1; }
  pid_t child = fork();   // before any HIP call
  g_me = child == 0 ? 1 : 0;
  g_sock = sv[g_me];
  close(sv[1 - g_me]);

  CHECK(hipSetDevice(0));
  int least = 0, greatest = 0;
  CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
  int canwait = -1;
  SOFT(hipDeviceGetAttribute(&canwait, hipDeviceAttributeCanUseStreamWaitValue, 0));
  if (g_me == 0) printf("stream priority range: least %d greatest %d; CanUseStreamWaitValue %d\n", least, greatest, canwait);
  hipStream_t s;
  CHECK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest));

  const size_t DATA = 64u << 20, PLANE = 8u << 20;
  char* data = nullptr;
  uint64_t* mbox = nullptr;
  uint32_t* err = nullptr;
  CHECK(hipMalloc(&data, DATA));
  CHECK(hipMalloc(&mbox, 4096));
  CHECK(hipMalloc(&err, 64));
  CHECK(hipMemset(mbox, 0, 4096));
  CHECK(hipMemset(err, 0, 64));
  k_fill<<<(DATA / 8 + 255) / 256, 256, 0, s>>>((uint64_t*)data, DATA / 8, (uint64_t)(g_me + 1) << 40);
  CHECK(hipStreamSynchronize(s));

  // ---- 1. memory handles: whole allocation, and a pointer 1 MiB inside it
  hipIpcMemHandle_t h_data, h_mid, h_mbox, p_data, p_mid, p_mbox;
  CHECK(hipIpcGetMemHandle(&h_data, data));
  bool mid_ok = SOFT(hipIpcGetMemHandle(&h_mid, data + (1u << 20)));
  CHECK(hipIpcGetMemHandle(&h_mbox, mbox));
  void* base = nullptr; size_t range = 0;
  if (SOFT(hipMemGetAddressRange((hipDeviceptr_t*)&base, &range, (hipDeviceptr_t)(data + (1u << 20)))) && g_me == 0)
    printf("hipMemGetAddressRange(data + 1 MiB): base offset %ld, size %zu MiB\n", (long)((char*)base - data), range >> 20);
  xsend(&h_data, sizeof h_data); xsend(&h_mid, sizeof h_mid); xsend(&h_mbox, sizeof h_mbox); xsend(&mid_ok, sizeof mid_ok);
  bool peer_mid_ok = false;
  xrecv(&p_data, sizeof p_data); xrecv(&p_mid, sizeof p_mid); xrecv(&p_mbox, sizeof p_mbox); xrecv(&peer_mid_ok, sizeof peer_mid_ok);
  if (g_me == 0) printf("handles of (data) and (data + 1 MiB) %s\n", memcmp(&h_data, &h_mid, sizeof h_data) ? "differ" : "are identical");
  char* rdata = nullptr; char* rmid = nullptr; uint64_t* rmbox = nullptr;
  CHECK(hipIpcOpenMemHandle((void**)&rdata, p_data, hipIpcMemLazyEnablePeerAccess));
  CHECK(hipIpcOpenMemHandle((void**)&rmbox, p_mbox, hipIpcMemLazyEnablePeerAccess));
  const uint64_t ptag = (uint64_t)(2 - g_me) << 40;   // the peer's fill tag
  uint64_t first = 0;
  if (peer_mid_ok && SOFT(hipIpcOpenMemHandle((void**)&rmid, p_mid, hipIpcMemLazyEnablePeerAccess))) {
    CHECK(hipMemcpy(&first, rmid, 8, hipMemcpyDeviceToHost));
    if (g_me == 0)
      printf("opening the handle of (data + 1 MiB) maps %s (first word %#lx, rmid - rdata = %ld)\n",
             first == ptag ? "the allocation's BASE" : first == ptag + (1u << 20) / 8 ? "the INNER pointer" : "something else",
             (unsigned long)first, (long)(rmid - rdata));
  }
  // opening the same handle a second time
  void* again = nullptr;
  bool twice = SOFT(hipIpcOpenMemHandle(&again, p_data, hipIpcMemLazyEnablePeerAccess));
  if (g_me == 0) printf("second hipIpcOpenMemHandle of the same handle: %s%s\n", twice ? "ok" : "refused", twice && again == rdata ? " (same address)" : "");
  host_barrier();

  // ---- 2. push a plane into the peer's buffer, flag kernel into the peer's mailbox, peer waits and checks
  // my planes [0, 8 MiB) go to the peer's [32 MiB, 40 MiB)
  CHECK(hipMemcpyAsync(rdata + (32u << 20), data, PLANE, hipMemcpyDeviceToDevice, s));
  k_signal<<<1, 1, 0, s>>>(rmbox + 0, 1);
  k_wait_ge<<<1, 1, 0, s>>>(mbox + 0, 1, 200000000ull, err);
  uint32_t* bad = err + 1;
  k_check<<<(PLANE / 8 + 255) / 256, 256, 0, s>>>((const uint64_t*)(data + (32u << 20)), PLANE / 8, ptag, bad);
  CHECK(hipStreamSynchronize(s));
  uint32_t herr[2];
  CHECK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost));
  printf("[p%d] push + flag: timeout %u, wrong words %u\n", g_me, herr[0], herr[1]);
  host_barrier();

  // ---- 3. copy rate into the peer mapping (same device here: a blit kernel; across devices SDMA over xGMI)
  {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) CHECK(hipMemcpyAsync(rdata + (40u << 20), data, PLANE, hipMemcpyDeviceToDevice, s));
    CHECK(hipEventRecord(e0, s));
    for (int w = 0; w < 20; ++w) CHECK(hipMemcpyAsync(rdata + (40u << 20), data, PLANE, hipMemcpyDeviceToDevice, s));
    CHECK(hipEventRecord(e1, s));
    CHECK(hipStreamSynchronize(s));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("[p%d] 8 MiB hipMemcpyAsync into the peer mapping: %.1f us each (%.0f GB/s)\n", g_me, ms * 50, PLANE / (ms / 20 * 1e-3) / 1e9);
  }
  host_barrier();

  // ---- 4. flag ping-pong with kernels: p0 signals n, p1 waits n and signals back
  {
    const int N = 2000;
    CHECK(hipMemset(err, 0, 64));
    host_barrier();
    const double t0 = now();
    for (int n = 1; n <= N; ++n) {
      if (g_me == 0) {
        k_signal<<<1, 1, 0, s>>>(rmbox + 8, (uint64_t)n);
        k_wait_ge<<<1, 1, 0, s>>>(mbox + 8, (uint64_t)n, 200000000ull, err);
      } else {
        k_wait_ge<<<1, 1, 0, s>>>(mbox + 8, (uint64_t)n, 200000000ull, err);
        k_signal<<<1, 1, 0, s>>>(rmbox + 8, (uint64_t)n);
      }
    }
    const double t_enq = now() - t0;
    CHECK(hipStreamSynchronize(s));
    const double t = now() - t0;
    CHECK(hipMemcpy(herr, err, 4, hipMemcpyDeviceToHost));
    printf("[p%d] kernel flag ping-pong: %.2f us per round trip (enqueue %.2f us), timeouts %u\n", g_me, t / N * 1e6, t_enq / N * 1e6, herr[0]);
  }
  host_barrier();

  // ---- 5. the same with stream memory operations (no kernel at all)
  if (canwait == 1) {
    const int N = 2000;
    bool ok = true;
    host_barrier();
    const double t0 = now();
    for (int n = 1; n <= N && ok; ++n) {
      if (g_me == 0) {
        ok = ok && SOFT(hipStreamWriteValue64(s, rmbox + 16, (uint64_t)n, 0));
        ok = ok && SOFT(hipStreamWaitValue64(s, mbox + 16, (uint64_t)n, hipStreamWaitValueGte, 0xffffffffffffffffull));
      } else {
        ok = ok && SOFT(hipStreamWaitValue64(s, mbox + 16, (uint64_t)n, hipStreamWaitValueGte, 0xffffffffffffffffull));
        ok = ok && SOFT(hipStreamWriteValue64(s, rmbox + 16, (uint64_t)n, 0));
      }
    }
    if (!ok) {   // release a peer that may be waiting
      k_signal<<<1, 1, 0, s>>>(rmbox + 16, (uint64_t)N + 1);
    }
    CHECK(hipStreamSynchronize(s));
    const double t = now() - t0;
    printf("[p%d] hipStreamWriteValue64/WaitValue64 ping-pong: %s, %.2f us per round trip\n", g_me, ok ? "ok" : "FAILED", t / N * 1e6);
  }
  host_barrier();

  // ---- 6. interprocess events: record here, the peer makes a stream wait for it
  {
    hipEvent_t ev = nullptr, pev = nullptr;
    hipIpcEventHandle_t he, pe;
    bool ok = SOFT(hipEventCreateWithFlags(&ev, hipEventInterprocess | hipEventDisableTiming));
    ok = ok && SOFT(hipIpcGetEventHandle(&he, ev));
    xsend(&ok, sizeof ok); if (ok) xsend(&he, sizeof he);
    bool pok = false; xrecv(&pok, sizeof pok); if (pok) xrecv(&pe, sizeof pe);
    bool both = ok && pok && SOFT(hipIpcOpenEventHandle(&pev, pe));
    char b = both; xsend(&b, 1); char pb = 0; xrecv(&pb, 1);
    both = both && pb;
    if (both) {
      const int N = 200;
      host_barrier();
      const double t0 = now();
      for (int n = 0; n < N; ++n) {
        // p0: record, tell p1; p1: wait for the record on its stream, sync, answer
        if (g_me == 0) {
          CHECK(hipEventRecord(ev, s));
          char c = 1; xsend(&c, 1); xrecv(&c, 1);
        } else {
          char c; xrecv(&c, 1);
          CHECK(hipStreamWaitEvent(s, pev, 0));
          CHECK(hipStreamSynchronize(s));
          xsend(&c, 1);
        }
      }
      printf("[p%d] interprocess event record -> peer stream wait + sync: %.1f us per round\n", g_me, (now() - t0) / N * 1e6);
    } else {
      printf("[p%d] interprocess events unavailable\n", g_me);
    }
  }
  host_barrier();
  SOFT(hipIpcCloseMemHandle(rdata));
  SOFT(hipIpcCloseMemHandle(rmbox));
  host_barrier();
  CHECK(hipFree(data)); CHECK(hipFree(mbox)); CHECK(hipFree(err));
  if (g_me == 0) {
    int st = 0;
    waitpid(child, &st, 0);
    printf("child exit %d\n", WIFEXITED(st) ? WEXITSTATUS(st) : -1);
    return WIFEXITED(st) ? WEXITSTATUS(st) : 1;
  }
  return 0;
}

// slab_peer.hpp -- the CU-free halo transport of include/neptune_hip.h section 8: ghost planes are contiguous, so a
// rank PUSHES its edge planes straight into its neighbour's ghost planes with hipMemcpyAsync (SDMA engines over xGMI
// between two devices; no CU takes part in moving the data) through a mapping of the neighbour's buffer
// (hipIpcGetMemHandle / hipIpcOpenMemHandle).  The reference has no counterpart (NeptunePETScRuntime.cpp:136:
// PETSC_COMM_SELF); SURVEY.md 8(e) defines the partitioning.
//
// What the ranks of one node share is a POSIX shared-memory segment named after the communicator id:
//   * per rank, host-side tables: the IPC handles of the device allocations it exchanges ("windows"), and per exchange
//     and side a target descriptor (window, offset, bytes) saying where its ghost planes are;
//   * per rank, a mailbox of four 64-bit counters that the neighbours' GPUs write and its own GPU polls.  Every rank
//     registers the segment with hipHostRegister, so the counters are fine-grained host memory visible to kernels of
//     all processes (and readable by the host for diagnostics).
// Interprocess HIP events cannot carry the handshake on this stack (hipStreamWaitEvent on an opened event handle is
// refused: tools/ipc_probe.hip, profiles/r03_ipc_probe.txt), so it is two one-wave kernels per exchange:
//
//   exchange n between a rank and its neighbour on side s (on the caller's stream, nothing blocks the host):
//     K1  tell the neighbour "my ghost planes on your side are free for push n" (everything enqueued before this
//         exchange on the stream -- the edge launches that read them -- is complete); wait for the same word from it
//     cp  hipMemcpyAsync: my edge planes -> the neighbour's ghost planes
//     K2  tell the neighbour "push n has landed"; wait for its push n to land here
//   Every device-side wait is bounded by the 100 MHz wall clock (NEPTUNE_HIP_PEER_TIMEOUT_S, default 20): a wave that
//   gives up sets an error word in host memory, which the next call on the communicator reports as NEPTUNE_HIP_ECOMM.
//   Host side, the neighbours' calls move in lock step within one exchange (a rank reads the neighbour's target
//   descriptor of exchange n, which exists once the neighbour has CALLED exchange n); those waits are bounded too.
#pragma once
#include <errno.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace neptune_hip {
namespace slab {
namespace peer {

constexpr int kMaxRanks = 16, kMaxWindows = 32;
constexpr int kRing = 16;   // target descriptors per side: a host runs at most one call (<= NEPTUNE_HIP_MAX_INPUTS fields) ahead of its neighbour
constexpr uint32_t kMagic = 0x4e505452u;  // "NPTR"
enum Side { LO = 0, HI = 1 };

struct Target {
  std::atomic<uint64_t> seq;   // exchange number this entry describes (written last)
  uint32_t window;
  uint64_t offset, bytes;
};
struct Window {
  hipIpcMemHandle_t handle;
  uint64_t bytes;
};
struct alignas(256) RankShm {
  // ---- device-visible mailbox: written by the neighbours' kernels, polled by this rank's
  alignas(64) uint64_t arrived[2];     // [s]: pushes of the neighbour on side s that have landed in my ghost planes
  alignas(64) uint64_t peer_free[2];   // [s]: the neighbour on side s accepts my pushes up to this exchange number
  // ---- host-side tables: written by this rank, read by its neighbours
  alignas(64) std::atomic<uint32_t> joined;
  std::atomic<uint32_t> n_windows;
  Window windows[kMaxWindows];
  Target target[2][kRing];             // [s]: where my ghost planes on side s are, per exchange
};
struct Shm {
  std::atomic<uint32_t> magic;
  uint32_t world;
  RankShm ranks[kMaxRanks];
};

inline double now_s() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
inline double timeout_s() {
  static const double t = [] {
    const char* e = getenv("NEPTUNE_HIP_PEER_TIMEOUT_S");
    const double v = (e && *e) ? atof(e) : 0.0;
    return v > 0 ? v : 20.0;
  }();
  return t;
}

struct State {
  Shm* shm = nullptr;         // host mapping of the segment
  Shm* dshm = nullptr;        // the same bytes as the device sees them (hipHostGetDevicePointer)
  bool registered = false;
  std::string name;
  int rank = 0, world = 1;
  uint64_t seq[2] = {0, 0};   // exchanges done with the neighbour on each side
  struct Win { char* base; size_t bytes; };
  std::vector<Win> windows;                         // my allocations, in publication order
  std::map<std::pair<int, int>, char*> mapped;      // (rank, window) -> base of my mapping of it
  uint32_t* err_host = nullptr;                     // host-mapped error word the wait kernels set on a timeout
  uint32_t* err_dev = nullptr;
  char error[256] = "";
};

inline std::string shm_name(const void* id) {
  const unsigned char* b = static_cast<const unsigned char*>(id);
  char buf[64] = "/neptune_hip_";
  size_t n = strlen(buf);
  for (int i = 0; i < 16; ++i) n += (size_t)snprintf(buf + n, sizeof buf - n, "%02x", b[i]);
  return buf;
}

inline void destroy(State* s) {
  if (!s) return;
  for (auto& m : s->mapped)
    if (m.first.first != s->rank && m.second) (void)hipIpcCloseMemHandle(m.second);
  if (s->err_host) (void)hipHostFree(s->err_host);
  if (s->registered) (void)hipHostUnregister(s->shm);
  if (s->shm) munmap(s->shm, sizeof(Shm));
  if (!s->name.empty()) shm_unlink(s->name.c_str());   // every rank tries; the name goes when the first one leaves
  (void)hipGetLastError();
  delete s;
}

// collective over the ranks of one node; nullptr (and `why`) on failure
inline State* create(const void* id, int rank, int world, char* why, size_t why_len) {
  auto fail = [&](State* s, const char* what, const char* detail) -> State* {
    snprintf(why, why_len, "%s: %s", what, detail ? detail : "?");
    destroy(s);
    return nullptr;
  };
  if (world > kMaxRanks) return fail(nullptr, "peer transport", "more ranks than one node holds");
  State* s = new State;
  s->rank = rank;
  s->world = world;
  s->name = shm_name(id);
  const double t0 = now_s();
  int fd = -1;
  if (rank == 0) {
    fd = shm_open(s->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return fail(s, "shm_open(create)", strerror(errno));
    if (ftruncate(fd, (off_t)sizeof(Shm)) != 0) { close(fd); return fail(s, "ftruncate", strerror(errno)); }
  } else {
    while ((fd = shm_open(s->name.c_str(), O_RDWR, 0600)) < 0) {
      if (now_s() - t0 > timeout_s()) { s->name.clear(); return fail(s, "shm_open", "rank 0's segment did not appear (ranks on different nodes?)"); }
      usleep(1000);
    }
    struct stat st;
    while (fstat(fd, &st) == 0 && (size_t)st.st_size < sizeof(Shm)) {
      if (now_s() - t0 > timeout_s()) { close(fd); s->name.clear(); return fail(s, "peer transport", "segment never sized"); }
      usleep(1000);
    }
  }
  void* p = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { if (rank != 0) s->name.clear(); return fail(s, "mmap", strerror(errno)); }
  s->shm = static_cast<Shm*>(p);
  if (rank == 0) {   // a fresh segment is zero-filled: every counter starts at 0
    s->shm->world = (uint32_t)world;
    s->shm->magic.store(kMagic, std::memory_order_release);
  } else {
    while (s->shm->magic.load(std::memory_order_acquire) != kMagic) {
      if (now_s() - t0 > timeout_s()) return fail(s, "peer transport", "segment never initialised");
      usleep(1000);
    }
    if ((int)s->shm->world != world) return fail(s, "peer transport", "ranks disagree on the world size");
  }
  if (hipHostRegister(s->shm, sizeof(Shm), hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) {
    (void)hipGetLastError();
    return fail(s, "hipHostRegister", "cannot make the shared segment visible to the device");
  }
  s->registered = true;
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, s->shm, 0) != hipSuccess) { (void)hipGetLastError(); return fail(s, "hipHostGetDevicePointer", "failed"); }
  s->dshm = static_cast<Shm*>(d);
  if (hipHostMalloc((void**)&s->err_host, 64, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return fail(s, "hipHostMalloc", "failed"); }
  *s->err_host = 0;
  void* de = nullptr;
  if (hipHostGetDevicePointer(&de, s->err_host, 0) != hipSuccess) { (void)hipGetLastError(); return fail(s, "hipHostGetDevicePointer", "failed"); }
  s->err_dev = static_cast<uint32_t*>(de);
  s->shm->ranks[rank].joined.store(1, std::memory_order_release);
  for (int r = 0; r < world; ++r)
    while (s->shm->ranks[r].joined.load(std::memory_order_acquire) != 1) {
      if (now_s() - t0 > timeout_s()) {
        char msg[96];
        snprintf(msg, sizeof msg, "rank %d never joined", r);
        return fail(s, "peer transport", msg);
      }
      usleep(500);
    }
  return s;
}

// my allocation holding [p, p + bytes): its window number (published on first use) and p's offset in it
inline int window_of(State* s, const void* p, size_t bytes, uint32_t* window, uint64_t* offset) {
  const char* c = static_cast<const char*>(p);
  for (size_t w = 0; w < s->windows.size(); ++w)
    if (c >= s->windows[w].base && c + bytes <= s->windows[w].base + s->windows[w].bytes) {
      *window = (uint32_t)w;
      *offset = (uint64_t)(c - s->windows[w].base);
      return 0;
    }
  if (s->windows.size() >= (size_t)kMaxWindows) { snprintf(s->error, sizeof s->error, "more than %d device allocations exchanged through one communicator", kMaxWindows); return -1; }
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) != hipSuccess) { (void)hipGetLastError(); snprintf(s->error, sizeof s->error, "hipMemGetAddressRange: not a device allocation"); return -1; }
  if (c + bytes > (char*)base + size) { snprintf(s->error, sizeof s->error, "field runs past its allocation"); return -1; }
  RankShm& me = s->shm->ranks[s->rank];
  const uint32_t w = (uint32_t)s->windows.size();
  // (the handle of an inner pointer opens at the allocation's base on this stack: export the base, ship the offset)
  if (hipIpcGetMemHandle(&me.windows[w].handle, base) != hipSuccess) { (void)hipGetLastError(); snprintf(s->error, sizeof s->error, "hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)"); return -1; }
  me.windows[w].bytes = size;
  s->windows.push_back({(char*)base, size});
  me.n_windows.store(w + 1, std::memory_order_release);
  *window = w;
  *offset = (uint64_t)(c - (char*)base);
  return 0;
}

// base of rank r's window w in this process
inline char* map_window(State* s, int r, uint32_t w) {
  if (r == s->rank) return w < s->windows.size() ? s->windows[w].base : nullptr;
  auto it = s->mapped.find({r, (int)w});
  if (it != s->mapped.end()) return it->second;
  RankShm& peer = s->shm->ranks[r];
  if (w >= peer.n_windows.load(std::memory_order_acquire)) { snprintf(s->error, sizeof s->error, "rank %d names a window it has not published", r); return nullptr; }
  void* p = nullptr;
  if (hipIpcOpenMemHandle(&p, peer.windows[w].handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
    (void)hipGetLastError();
    snprintf(s->error, sizeof s->error, "hipIpcOpenMemHandle of rank %d's buffer failed", r);
    return nullptr;
  }
  s->mapped[{r, (int)w}] = static_cast<char*>(p);
  return static_cast<char*>(p);
}

}  // namespace peer
}  // namespace slab
}  // namespace neptune_hip

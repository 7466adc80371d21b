// wisdom.hip -- include/neptune_hip.h "launch wisdom": measured launch choices kept in a text file next to the module
// cache, plus the per-thread record of the last launch.  Host code only (its own translation unit: builds in seconds,
// linked into libneptune_hip.so).  The reference has nothing to tune (DataflowLowering.cpp:289-310 is one loop nest).
//
// File format, one line per choice, appended with a single write() (O_APPEND: concurrent processes do not tear lines):
//     <device>|<key> \t <kernel> <variant> <chunk> <flags> <ms>
// Unknown or malformed lines are ignored; the last line of a key wins.
#include <errno.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <map>
#include <mutex>
#include <string>

#include "../../../include/neptune_hip.h"

namespace {

struct Wisdom {
  std::mutex mu;
  bool loaded = false;
  std::string path;          // "" = disabled
  std::string device;        // "<name>/<arch>/<CUs>" of the current device
  std::map<std::string, neptune_hip_launch_cfg_t> entries;
};
Wisdom& W() {
  static Wisdom w;
  return w;
}
std::atomic<long long> g_measured{0}, g_hits{0}, g_stored{0};

void mkdirs(const std::string& dir) {
  for (size_t i = 1; i <= dir.size(); ++i)
    if (i == dir.size() || dir[i] == '/') (void)mkdir(dir.substr(0, i).c_str(), 0755);
}

// caller holds w.mu
void load(Wisdom& w) {
  if (w.loaded) return;
  w.loaded = true;
  const char* e = getenv("NEPTUNE_HIP_WISDOM");
  if (e) {
    w.path = e;   // empty = disabled
  } else {
    const char* c = getenv("NEPTUNE_CACHE_DIR");
    const char* h = getenv("HOME");
    std::string dir = (c && *c) ? std::string(c) : (std::string((h && *h) ? h : "/tmp") + "/.neptune/cache");
    mkdirs(dir);
    w.path = dir + "/wisdom_v1.txt";
  }
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) {
    w.device = std::string(prop.name) + "/" + prop.gcnArchName + "/" + std::to_string(prop.multiProcessorCount);
    for (char& ch : w.device)
      if (ch == '\t' || ch == '\n' || ch == '|') ch = '_';
  } else {
    (void)hipGetLastError();
    w.device = "unknown-device";
  }
  if (w.path.empty()) return;
  FILE* f = fopen(w.path.c_str(), "r");
  if (!f) return;
  char* line = nullptr;
  size_t cap = 0;
  while (getline(&line, &cap, f) > 0) {
    char* tab = strchr(line, '\t');
    if (!tab) continue;
    *tab = 0;
    neptune_hip_launch_cfg_t c{};
    double ms = 0;
    if (sscanf(tab + 1, "%d %d %d %d %lf", &c.kernel, &c.variant, &c.chunk, &c.flags, &ms) != 5) continue;
    if (c.kernel < 0 || c.kernel > NEPTUNE_HIP_KERNEL_MARCH || c.chunk < 0) continue;
    w.entries[line] = c;
  }
  free(line);
  fclose(f);
}

thread_local neptune_hip_launch_cfg_t t_last = {-1, -1, 0, 0};

}  // namespace

extern "C" {

int neptune_hip_wisdom_lookup(const char* key, neptune_hip_launch_cfg_t* cfg) {
  if (!key || !cfg) return 0;
  Wisdom& w = W();
  std::lock_guard<std::mutex> lk(w.mu);
  load(w);
  auto it = w.entries.find(w.device + "|" + key);
  if (it == w.entries.end()) return 0;
  *cfg = it->second;
  ++g_hits;
  return 1;
}

int neptune_hip_wisdom_store(const char* key, const neptune_hip_launch_cfg_t* cfg, double ms) {
  if (!key || !cfg || strchr(key, '\t') || strchr(key, '\n')) return NEPTUNE_HIP_EINVAL;
  Wisdom& w = W();
  std::lock_guard<std::mutex> lk(w.mu);
  load(w);
  ++g_measured;
  const std::string full = w.device + "|" + key;
  w.entries[full] = *cfg;
  if (w.path.empty()) return NEPTUNE_HIP_OK;
  char tail[96];
  snprintf(tail, sizeof tail, "\t%d %d %d %d %.6f\n", cfg->kernel, cfg->variant, cfg->chunk, cfg->flags, ms);
  const std::string line = full + tail;
  const int fd = open(w.path.c_str(), O_WRONLY | O_CREAT | O_APPEND, 0644);
  if (fd < 0) return NEPTUNE_HIP_OK;   // a read-only cache directory costs the persistence, not the launch
  const ssize_t n = write(fd, line.data(), line.size());
  close(fd);
  if (n == (ssize_t)line.size()) ++g_stored;
  return NEPTUNE_HIP_OK;
}

const char* neptune_hip_wisdom_path(void) {
  Wisdom& w = W();
  std::lock_guard<std::mutex> lk(w.mu);
  load(w);
  return w.path.c_str();
}

void neptune_hip_tune_stats(int64_t out[3]) {
  if (!out) return;
  out[0] = g_measured.load();
  out[1] = g_hits.load();
  out[2] = g_stored.load();
}

void neptune_hip_note_launch(int kernel, int variant, int chunk) { t_last = {kernel, variant, chunk, 0}; }

int neptune_hip_last_launch(neptune_hip_launch_cfg_t* out) {
  if (!out || t_last.kernel < 0) return 0;
  *out = t_last;
  return 1;
}

}  // extern "C"

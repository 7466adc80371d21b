// capi.cpp -- extern "C" entry points of libneptune_lowering.so (bound by the Python frontend
// through ctypes; declared for C callers in include/neptune_lowering.h).
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <exception>
#include <filesystem>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include <fcntl.h>
#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>

#include "lowering.h"

extern char** environ;

using namespace neptune_lowering;

namespace {
char* dup(const std::string& s) {
  char* p = (char*)std::malloc(s.size() + 1);
  if (p) std::memcpy(p, s.c_str(), s.size() + 1);
  return p;
}
std::string report_json(const LowerInfo& info) {
  std::ostringstream o;
  o << "{\"lowered\": [";
  for (size_t i = 0; i < info.lowered.size(); ++i) o << (i ? ", " : "") << "\"" << info.lowered[i] << "\"";
  o << "], \"signatures\": [";
  auto sigtype = [&](const SigType& t) {
    o << "{\"kind\": \"" << t.kind << "\", \"elem\": \"" << t.elem << "\", \"rank\": " << t.rank << ", \"shape\": [";
    for (size_t i = 0; i < t.shape.size(); ++i) o << (i ? ", " : "") << t.shape[i];
    o << "], \"lb\": [";
    for (size_t i = 0; i < t.lb.size(); ++i) o << (i ? ", " : "") << t.lb[i];
    o << "]";
    if (!t.scalar.empty()) o << ", \"scalar\": \"" << t.scalar << "\"";
    o << "}";
  };
  for (size_t i = 0; i < info.signatures.size(); ++i) {
    const Signature& s = info.signatures[i];
    o << (i ? ", " : "") << "{\"name\": \"" << s.name << "\", \"args\": [";
    for (size_t k = 0; k < s.args.size(); ++k) { if (k) o << ", "; sigtype(s.args[k]); }
    o << "], \"result\": ";
    if (s.has_result) sigtype(s.result); else o << "null";
    o << "}";
  }
  o << "], \"skipped\": [";
  for (size_t i = 0; i < info.skipped.size(); ++i) {
    std::string why = info.skipped[i].second;
    for (auto& c : why) if (c == '"') c = '\'';
    o << (i ? ", " : "") << "{\"symbol\": \"" << info.skipped[i].first << "\", \"reason\": \"" << why << "\"}";
  }
  o << "], \"outlined\": [";
  for (size_t i = 0; i < info.outlined.size(); ++i) {
    const auto& x = info.outlined[i];
    o << (i ? ", " : "") << "{\"symbol\": \"" << x.symbol << "\", \"function\": \"" << x.function << "\", \"value\": \"" << x.value
      << "\", \"line\": " << x.line << "}";
  }
  o << "], \"applies\": [";
  for (size_t i = 0; i < info.applies.size(); ++i) {
    const ApplyInfo& a = info.applies[i];
    o << (i ? ", " : "") << "{\"function\": \"" << a.function << "\", \"tag\": \"" << a.tag << "\", \"rank\": " << a.rank
      << ", \"inputs\": " << a.num_inputs << ", \"kernel\": \"" << (a.fused_reduce ? "reduce" : (a.march ? "march" : "direct"))
      << "\", \"shape\": \""
      << (a.halo_input < 0 ? "pointwise" : (a.box ? "box" : "star")) << "\", \"elem\": \"" << a.elem << "\", \"halo0\": " << a.halo0
      << ", \"geom_symbol\": \"" << a.geom_symbol << "\", \"exact\": " << (a.exact ? "true" : "false") << "}";
  }
  o << "]}";
  return o.str();
}
// Run `argv` as a child process with stdout/stderr in `log_path` and wait for it.  The child's environment is the
// caller's minus everything a profiler or tool injects into a process (LD_PRELOAD, ROCP* / ROCPROFILER* / HSA_TOOLS_*):
// under `rocprofv3 --pmc` the preloaded library initialises the GPU in every process that inherits it, and hipcc execs
// clang, lld, ... in turn -- a chain of exec hops from GPU-initialised processes, which this pool forbids.  The compiler
// needs none of those variables.  Returns the exit status, or -1 (why in `err`).
bool tool_variable(const char* kv) {
  static const char* const prefixes[] = {"LD_PRELOAD=", "ROCP", "ROCPROFILER", "HSA_TOOLS_", "ROCTX", "ROCTRACER", "RPD_"};
  for (const char* p : prefixes)
    if (std::strncmp(kv, p, std::strlen(p)) == 0) return true;
  return false;
}
int run_tool(const std::vector<std::string>& argv, const std::string& log_path, std::string& err) {
  std::vector<char*> av;
  for (const std::string& a : argv) av.push_back(const_cast<char*>(a.c_str()));
  av.push_back(nullptr);
  std::vector<char*> ev;
  for (char** e = environ; e && *e; ++e)
    if (!tool_variable(*e)) ev.push_back(*e);
  ev.push_back(nullptr);
  posix_spawn_file_actions_t fa;
  posix_spawn_file_actions_init(&fa);
  posix_spawn_file_actions_addopen(&fa, 1, log_path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
  posix_spawn_file_actions_adddup2(&fa, 1, 2);
  posix_spawn_file_actions_addopen(&fa, 0, "/dev/null", O_RDONLY, 0);
  pid_t pid = 0;
  const int rc = posix_spawn(&pid, av[0], &fa, nullptr, av.data(), ev.data());
  posix_spawn_file_actions_destroy(&fa);
  if (rc != 0) { err = std::string("cannot start ") + av[0] + ": " + std::strerror(rc); return -1; }
  int status = 0;
  while (waitpid(pid, &status, 0) < 0) {
    if (errno != EINTR) { err = std::string("waitpid: ") + std::strerror(errno); return -1; }
  }
  if (WIFEXITED(status)) return WEXITSTATUS(status);
  err = "terminated by signal " + std::to_string(WIFSIGNALED(status) ? WTERMSIG(status) : 0);
  return -1;
}
// names the kernel sources a module is compiled from (FNV-1a over the headers, in name order): part of every
// launch-wisdom key of the module (include/neptune_hip.h "launch wisdom"), so choices measured against other kernels
// are never reused
std::string kernel_build_id(const std::string& root) {
  namespace fs = std::filesystem;
  std::vector<std::string> files;
  std::error_code ec;
  for (const char* sub : {"/neptune-pde-solver_amd/csrc/kernels", "/neptune-pde-solver_amd/csrc/runtime"})
    for (fs::directory_iterator it(root + sub, ec), end; !ec && it != end; it.increment(ec))
      if (it->path().extension() == ".hpp") files.push_back(it->path().string());
  files.push_back(root + "/include/neptune_hip.h");
  std::sort(files.begin(), files.end());
  unsigned long long h = 1469598103934665603ull;
  for (const std::string& f : files) {
    std::ifstream in(f, std::ios::binary);
    char buf[65536];
    while (in.read(buf, sizeof buf) || in.gcount() > 0)
      for (std::streamsize i = 0; i < in.gcount(); ++i) { h ^= (unsigned char)buf[i]; h *= 1099511628211ull; }
  }
  char out[32];
  std::snprintf(out, sizeof out, "%016llx", h);
  return out;
}
bool front(const char* text, Module& m, Diag& d) {
  if (!text) { d.fail(0, "null module text"); return false; }
  return parse_module(text, m, d) && verify_module(m, d);
}
// Nothing may leave the C ABI as a C++ exception: malformed text that slips past a structural check (an op with
// fewer types or operands than its kind implies) surfaces as std::out_of_range from a checked access -- report
// it as a diagnostic like any other rejection.
template <class F>
int guarded(char** diag_out, F&& body) {
  try {
    return body();
  } catch (const std::exception& e) {
    if (diag_out) *diag_out = dup(std::string("malformed module (") + e.what() + ")");
    return -1;
  } catch (...) {
    if (diag_out) *diag_out = dup("malformed module");
    return -1;
  }
}
}  // namespace

extern "C" {

const char* neptune_lowering_version(void) { return "neptune-lowering 0.1 (NeptuneIR hot-path subset -> HIP, gfx950)"; }

void neptune_lowering_free(char* p) { std::free(p); }

int neptune_lowering_verify(const char* mlir_text, char** diag_out) {
  if (diag_out) *diag_out = nullptr;
  return guarded(diag_out, [&] {
    Module m;
    Diag d;
    const bool ok = front(mlir_text, m, d);
    if (diag_out) *diag_out = ok ? nullptr : dup(d.message);
    return ok ? 0 : -1;
  });
}

int neptune_lowering_to_hip(const char* mlir_text, char** source_out, char** report_out, char** diag_out) {
  if (source_out) *source_out = nullptr;
  if (report_out) *report_out = nullptr;
  if (diag_out) *diag_out = nullptr;
  return guarded(diag_out, [&] {
    Module m;
    Diag d;
    std::string src;
    LowerInfo info;
    if (!front(mlir_text, m, d) || !lower_to_hip(m, src, info, d)) {
      if (diag_out) *diag_out = dup(d.message);
      return -1;
    }
    if (source_out) *source_out = dup(src);
    if (report_out) *report_out = dup(report_json(info));
    return 0;
  });
}

int neptune_lowering_compile(const char* mlir_text, const char* so_path, const char* repo_root, const char* hipcc,
                             char** report_out, char** diag_out) {
  char* src = nullptr;
  int rc = neptune_lowering_to_hip(mlir_text, &src, report_out, diag_out);
  if (rc != 0) return rc;
  if (!so_path || !repo_root) { if (diag_out) *diag_out = dup("so_path and repo_root are required"); std::free(src); return -1; }
  const std::string hip_path = std::string(so_path) + ".hip";
  {
    std::ofstream f(hip_path);
    if (!f) { if (diag_out) *diag_out = dup("cannot write " + hip_path); std::free(src); return -1; }
    f << src;
  }
  std::free(src);
  const std::string cc = (hipcc && *hipcc) ? hipcc : "/opt/rocm/bin/hipcc";
  const std::string root = repo_root;
  const std::string libdir = root + "/neptune-pde-solver_amd/lib";
  const std::string log = std::string(so_path) + ".log";
  // -ffp-contract=off: bodies must evaluate op by op like the reference's FMA-free lowering.
  // NEPTUNE_HIP_FULL_VARIANTS=1 compiles every march tile into the module (longer build; for NEPTUNE_HIP_TUNE=1).
  const char* fullv = std::getenv("NEPTUNE_HIP_FULL_VARIANTS");
  std::vector<std::string> argv = {cc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"};
  if (fullv && *fullv && *fullv != '0') argv.push_back("-DNEPTUNE_HIP_FULL_VARIANTS=1");
  argv.push_back("-DNEPTUNE_HIP_BUILD_ID=\"" + kernel_build_id(root) + "\"");
  for (const std::string& a : {std::string("-x"), std::string("hip"), hip_path, "-I" + root, "-L" + libdir, std::string("-lneptune_hip"),
                               "-Wl,-rpath," + libdir, std::string("-o"), std::string(so_path)})
    argv.push_back(a);
  std::string why;
  rc = run_tool(argv, log, why);
  if (rc != 0) {
    std::ifstream f(log);
    std::stringstream ss;
    ss << "hipcc failed (";
    for (size_t i = 0; i < argv.size(); ++i) ss << (i ? " " : "") << argv[i];
    ss << ")" << (why.empty() ? "" : ": " + why) << ":\n" << f.rdbuf();
    if (diag_out) *diag_out = dup(ss.str());
    return -2;
  }
  return 0;
}

}  // extern "C"

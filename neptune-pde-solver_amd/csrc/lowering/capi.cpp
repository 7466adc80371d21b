// capi.cpp -- extern "C" entry points of libneptune_lowering.so (bound by the Python frontend
// through ctypes; declared for C callers in include/neptune_lowering.h).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <sstream>

#include "lowering.h"

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
  const std::string defs = (fullv && *fullv && *fullv != '0') ? " -DNEPTUNE_HIP_FULL_VARIANTS=1" : "";
  const std::string cmd = cc + " --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared" + defs + " -x hip '" + hip_path +
                          "' -I'" + root + "' -L'" + libdir + "' -lneptune_hip -Wl,-rpath,'" + libdir + "' -o '" + so_path +
                          "' > '" + log + "' 2>&1";
  rc = std::system(cmd.c_str());
  if (rc != 0) {
    std::ifstream f(log);
    std::stringstream ss;
    ss << "hipcc failed (" << cmd << "):\n" << f.rdbuf();
    if (diag_out) *diag_out = dup(ss.str());
    return -2;
  }
  return 0;
}

}  // extern "C"

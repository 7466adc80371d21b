// neptune-opt -- drop-in for the reference's `neptune-opt` driver (src/neptuneOpt.cpp:38-47) on
// the stencil hot path.  The reference tool is MlirOptMain + its pipelines and is used as
//     neptune-opt in.mlir --neptuneir-to-llvm            (README.md:60-64, smoke_apply.sh:21-22)
// This tool accepts the same input files and offers the HIP pipeline instead:
//     neptune-opt in.mlir --neptuneir-to-hip [-o out.hip]        emit the HIP translation unit
//     neptune-opt in.mlir --neptuneir-to-hip --emit=so -o out.so emit and compile (hipcc, gfx950)
//     neptune-opt in.mlir --verify-only                           run the verifiers only
//     neptune-opt in.mlir --report                                 print what was lowered, as JSON
// --neptuneir-to-llvm is recognised and answered with a pointer to --neptuneir-to-hip (there is
// no MLIR/LLVM in this toolchain; the CPU pipeline stays with the reference build).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>

#include "lowering.h"

static std::string self_repo_root(const char* argv0) {
  // <repo>/neptune-pde-solver_amd/bin/neptune-opt -> <repo>
  char buf[4096];
  if (!realpath(argv0, buf)) return ".";
  std::string p = buf;
  for (int i = 0; i < 3; ++i) {
    size_t s = p.find_last_of('/');
    if (s == std::string::npos) return ".";
    p = p.substr(0, s);
  }
  return p;
}

int main(int argc, char** argv) {
  std::string in, out, emit = "hip";
  bool to_hip = false, verify_only = false, report = false, to_llvm = false;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "--neptuneir-to-hip") to_hip = true;
    else if (a == "--neptuneir-to-llvm") to_llvm = true;
    else if (a == "--verify-only" || a == "--neptune-ir-verify-annotate") verify_only = true;
    else if (a == "--report") report = true;
    else if (a.compare(0, 7, "--emit=") == 0) emit = a.substr(7);
    else if (a == "-o" && i + 1 < argc) out = argv[++i];
    else if (a == "--version") { std::puts(neptune_lowering_version()); return 0; }
    else if (a == "-h" || a == "--help") {
      std::puts("usage: neptune-opt <in.mlir|-> (--neptuneir-to-hip [--emit=hip|so] [-o file] | --verify-only) [--report]");
      return 0;
    } else if (!a.empty() && a[0] == '-' && a != "-") { std::fprintf(stderr, "neptune-opt: unknown option %s\n", a.c_str()); return 2; }
    else in = a;
  }
  if (to_llvm) {
    std::fprintf(stderr, "neptune-opt: --neptuneir-to-llvm is the reference's CPU pipeline (needs MLIR/LLVM); "
                         "this build provides the HIP pipeline: use --neptuneir-to-hip\n");
    return 2;
  }
  if (in.empty()) { std::fprintf(stderr, "neptune-opt: no input file\n"); return 2; }
  std::stringstream ss;
  if (in == "-") ss << std::cin.rdbuf();
  else {
    std::ifstream f(in);
    if (!f) { std::fprintf(stderr, "neptune-opt: cannot open %s\n", in.c_str()); return 2; }
    ss << f.rdbuf();
  }
  const std::string text = ss.str();
  char *diag = nullptr, *src = nullptr, *rep = nullptr;
  struct Release {  // the library hands out malloc'ed strings (include/neptune_lowering.h)
    char *&a, *&b, *&c;
    ~Release() { neptune_lowering_free(a); neptune_lowering_free(b); neptune_lowering_free(c); }
  } release{diag, src, rep};
  if (verify_only && !to_hip) {
    if (neptune_lowering_verify(text.c_str(), &diag) != 0) { std::fprintf(stderr, "%s: error: %s\n", in.c_str(), diag); return 1; }
    return 0;
  }
  if (!to_hip) { std::fprintf(stderr, "neptune-opt: nothing to do (pass --neptuneir-to-hip or --verify-only)\n"); return 2; }
  if (emit == "so") {
    if (out.empty()) { std::fprintf(stderr, "neptune-opt: --emit=so needs -o <file.so>\n"); return 2; }
    const char* root_env = std::getenv("NEPTUNE_HIP_ROOT");
    const std::string root = root_env ? root_env : self_repo_root(argv[0]);
    int rc = neptune_lowering_compile(text.c_str(), out.c_str(), root.c_str(), std::getenv("HIPCC"), &rep, &diag);
    if (rc != 0) { std::fprintf(stderr, "%s: error: %s\n", in.c_str(), diag ? diag : "?"); return 1; }
    if (report && rep) std::puts(rep);
    return 0;
  }
  if (neptune_lowering_to_hip(text.c_str(), &src, &rep, &diag) != 0) {
    std::fprintf(stderr, "%s: error: %s\n", in.c_str(), diag ? diag : "?");
    return 1;
  }
  if (out.empty()) std::fputs(src, stdout);
  else {
    std::ofstream f(out);
    f << src;
  }
  if (report && rep) std::fprintf(out.empty() ? stderr : stdout, "%s\n", rep);
  return 0;
}

// lowering.h -- public face of the NeptuneIR -> HIP lowering library.
#pragma once
#include <string>
#include <utility>
#include <vector>

#include "ir.h"

namespace neptune_lowering {

struct ApplyInfo {
  std::string function, tag;
  int rank = 0, num_inputs = 0, halo_input = -1;
  bool march = false, box = false;
  bool fused_reduce = false;  // evaluated inside the consuming reduce's kernel
  std::string elem;           // element type of the apply
  int halo0 = 0;              // reach along dim 0 (what a slab decomposition must hold as ghost planes)
  std::string geom_symbol;    // exported geometry-level entry (empty: none)
  bool exact = true;          // false: the body uses elementary functions (exp, log, ...): a few ulp, not bit-exact
};
struct SigType {
  std::string kind, elem;  // kind: memref | temp | field
  int rank = 0;
  std::vector<int64_t> shape;  // -1 = dynamic
  std::vector<int64_t> lb;     // temp / field: logical origin
  // scalar results only: what the value is when the function runs on one slab of a decomposed field --
  // "uniform" (the same on every rank), "partial_sum" (a bare reduce: the ranks' values add up) or "derived"
  // (computed from a partial sum: only right on a single rank)
  std::string scalar;
};
struct Signature {
  std::string name;
  std::vector<SigType> args;
  bool has_result = false;
  SigType result;
};
struct LowerInfo {
  std::vector<Signature> signatures;                           // one per exported symbol
  std::vector<std::string> lowered;                            // exported symbols
  std::vector<std::pair<std::string, std::string>> skipped;    // (symbol, reason)
  std::vector<ApplyInfo> applies;
  // stencil parts of functions that are not lowered as a whole (they hold solver ops): each is an exported symbol that
  // computes one value the solver op consumes, from the function's own arguments
  struct Outlined { std::string symbol, function, value; int line = 0; };
  std::vector<Outlined> outlined;
};

// emit_hip.cpp
bool lower_to_hip(const Module& m, std::string& out_source, LowerInfo& info, Diag& diag);

}  // namespace neptune_lowering

// ---- C ABI (include/neptune_lowering.h documents it) ---------------------------------------
extern "C" {
int neptune_lowering_verify(const char* mlir_text, char** diag_out);
int neptune_lowering_to_hip(const char* mlir_text, char** source_out, char** report_out, char** diag_out);
int neptune_lowering_compile(const char* mlir_text, const char* so_path, const char* repo_root, const char* hipcc,
                             char** report_out, char** diag_out);
void neptune_lowering_free(char* p);
const char* neptune_lowering_version(void);
}

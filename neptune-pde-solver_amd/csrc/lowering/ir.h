// ir.h -- in-memory form of the NeptuneIR hot-path subset accepted by the HIP lowering.
//
// The reference builds on MLIR (include/Dialect/NeptuneIR/*.td); neither MLIR nor LLVM exists
// in this toolchain, so the lowering carries its own small IR and a parser for the textual
// assembly formats the reference's ODS declares:
//   wrap / unwrap / load            NeptuneIROps.td:20-84
//   apply / access / yield / store  NeptuneIROps.td:94-259
//   linear_opdef / nonlinear_opdef / apply_linear / apply_nonlinear / return
//                                   NeptuneIROps.td:124-131, 318-520
//   !neptune_ir.field / temp, #neptune_ir.bounds / location
//                                   NeptuneIRTypes.td:12-59, NeptuneIRAttrs.td:9-49
// plus the arith / math / scf ops that appear inside apply regions and func.func.
// Ops outside that subset (time_advance, assemble_matrix, solve_*, reduce, ...) are parsed as
// opaque so the stencil operators around them can still be lowered; a function that contains
// one is reported as "not lowered" (the PETSc solver path stays on the host, unchanged).
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace neptune_lowering {

struct Bounds {
  std::vector<int64_t> lb, ub;
  int rank() const { return (int)lb.size(); }
  bool operator==(const Bounds& o) const { return lb == o.lb && ub == o.ub; }
};

enum class TypeKind { Scalar, Temp, Field, MemRef, None };

struct Type {
  TypeKind kind = TypeKind::None;
  std::string elem;            // f64 f32 index i1 i32 i64 (scalar name or element type)
  Bounds bounds;               // Temp / Field
  std::string location;        // Temp / Field
  std::vector<int64_t> shape;  // MemRef; -1 = dynamic ('?')
  bool tensor = false;         // MemRef spelled `tensor<...>` (as_tensor / from_tensor): a dense buffer all the same
  bool is_scalar() const { return kind == TypeKind::Scalar; }
  bool is_tempish() const { return kind == TypeKind::Temp || kind == TypeKind::Field; }
  int rank() const { return kind == TypeKind::MemRef ? (int)shape.size() : bounds.rank(); }
  bool operator==(const Type& o) const {
    return kind == o.kind && elem == o.elem && bounds == o.bounds && location == o.location && shape == o.shape && tensor == o.tensor;
  }
  bool operator!=(const Type& o) const { return !(*this == o); }
  std::string str() const;
};

struct AttrValue {
  enum Kind { None, Int, Float, String, Symbol, BoundsK, Bool, Unit } kind = None;
  int64_t i = 0;
  double f = 0;
  std::string s;   // String / Symbol / literal text of numbers
  Bounds bounds;
  bool b = false;
};

struct Block;

struct Op {
  std::string name;                   // e.g. "neptune_ir.apply", "arith.addf"
  std::vector<std::string> results;   // SSA names including '%'
  std::vector<std::string> operands;
  std::map<std::string, AttrValue> attrs;
  std::vector<Type> types;            // op-specific; see parser.cpp
  std::vector<std::unique_ptr<Block>> regions;
  std::vector<int64_t> offsets;       // neptune_ir.access
  std::string callee;                 // apply_linear / apply_nonlinear
  std::string literal;                // arith.constant: literal text
  std::string predicate;              // arith.cmpi / cmpf
  bool opaque = false;                // parsed by skipping: outside the hot path
  int line = 0;
};

struct BlockArg {
  std::string name;
  Type type;
};

struct Block {
  std::vector<BlockArg> args;
  std::vector<std::unique_ptr<Op>> ops;
};

enum class FuncKind { Func, LinearOpDef, NonlinearOpDef };

struct Function {
  std::string name;
  FuncKind kind = FuncKind::Func;
  std::vector<Type> arg_types;
  std::vector<Type> result_types;
  Block body;
  int line = 0;
};

struct Module {
  std::vector<std::unique_ptr<Function>> funcs;
  Function* find(const std::string& name) const {
    for (auto& f : funcs)
      if (f->name == name) return f.get();
    return nullptr;
  }
};

struct Diag {
  bool ok = true;
  std::string message;  // first error, "line N: ..." form
  void fail(int line, const std::string& msg) {
    if (!ok) return;
    ok = false;
    message = (line > 0 ? "line " + std::to_string(line) + ": " : "") + msg;
  }
};

// parser.cpp
bool parse_module(const std::string& text, Module& out, Diag& diag);
// verify.cpp: the checks of ApplyOp::verify (lib/Dialect/NeptuneIR/NeptuneIRVerifier.cpp:141-171),
// checkApplyLike (lib/Passes/VerifyAndAnnotate.cpp:87-214) and the linear_opdef body whitelist
// (NeptuneIRVerifier.cpp:34-118), with the reference's diagnostics.
bool verify_module(const Module& m, Diag& diag);

}  // namespace neptune_lowering

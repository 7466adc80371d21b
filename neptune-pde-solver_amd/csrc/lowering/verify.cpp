// verify.cpp -- structural checks the reference performs before lowering, with its diagnostics:
//   ApplyOp::verify                 lib/Dialect/NeptuneIR/NeptuneIRVerifier.cpp:141-171
//   verifyLinearOpBody / ...Region  lib/Dialect/NeptuneIR/NeptuneIRVerifier.cpp:34-118
//   checkApplyLike                  lib/Passes/VerifyAndAnnotate.cpp:87-214
// plus the SSA/type consistency MLIR's own verifier would have enforced on the way in.
#include <set>

#include "ir.h"

namespace neptune_lowering {
namespace {

using Scope = std::map<std::string, Type>;

Type scalar(const std::string& e) {
  Type t;
  t.kind = TypeKind::Scalar;
  t.elem = e;
  return t;
}
bool is_float(const Type& t) { return t.is_scalar() && (t.elem == "f64" || t.elem == "f32"); }
bool is_intlike(const Type& t) { return t.is_scalar() && (t.elem == "index" || t.elem == "i64" || t.elem == "i32" || t.elem == "i1"); }

struct Verifier {
  const Module& m;
  Diag& diag;
  Verifier(const Module& mm, Diag& d) : m(mm), diag(d) {}

  bool lookup(const Scope& sc, const Op& op, const std::string& v, Type& out) {
    auto it = sc.find(v);
    if (it == sc.end()) { diag.fail(op.line, "use of undefined value " + v + " in '" + op.name + "'"); return false; }
    out = it->second;
    return true;
  }

  // arith / math ops on scalars: operand types must match the declared type
  bool check_arith(const Op& op, Scope& sc) {
    const std::string& n = op.name;
    std::vector<Type> ots;
    for (auto& v : op.operands) { Type t; if (!lookup(sc, op, v, t)) return false; ots.push_back(t); }
    auto need = [&](size_t k) { if (ots.size() != k) { diag.fail(op.line, "'" + n + "' expects " + std::to_string(k) + " operands"); return false; } return true; };
    const Type decl = op.types.empty() ? Type{} : op.types[0];
    static const std::set<std::string> binf = {"arith.addf", "arith.subf", "arith.mulf", "arith.divf", "arith.maximumf",
                                               "arith.minimumf", "arith.maxnumf", "arith.minnumf", "math.powf",
                                               "math.copysign"};
    static const std::set<std::string> bini = {"arith.addi", "arith.subi", "arith.muli", "arith.andi", "arith.ori", "arith.xori"};
    // math.exp ... math.tanh: elementary functions (libm / device math library: not correctly rounded, a few ulp)
    static const std::set<std::string> unf = {"arith.negf", "math.sqrt", "math.absf", "math.floor", "math.ceil", "math.exp",
                                              "math.log", "math.sin", "math.cos", "math.tanh"};
    if (binf.count(n)) {
      if (!need(2)) return false;
      if (!is_float(decl) || ots[0] != decl || ots[1] != decl) { diag.fail(op.line, "'" + n + "' operand types must equal its floating-point type"); return false; }
      sc[op.results.at(0)] = decl;
    } else if (bini.count(n)) {
      if (!need(2)) return false;
      if (!is_intlike(decl) || ots[0] != decl || ots[1] != decl) { diag.fail(op.line, "'" + n + "' operand types must equal its integer type"); return false; }
      sc[op.results.at(0)] = decl;
    } else if (unf.count(n)) {
      if (!need(1)) return false;
      if (!is_float(decl) || ots[0] != decl) { diag.fail(op.line, "'" + n + "' operand type must equal its floating-point type"); return false; }
      sc[op.results.at(0)] = decl;
    } else if (n == "arith.cmpi" || n == "arith.cmpf") {
      if (!need(2)) return false;
      const bool f = n == "arith.cmpf";
      if ((f ? !is_float(decl) : !is_intlike(decl)) || ots[0] != decl || ots[1] != decl) { diag.fail(op.line, "'" + n + "' operand types must equal its declared type"); return false; }
      static const std::set<std::string> pi = {"eq", "ne", "slt", "sle", "sgt", "sge", "ult", "ule", "ugt", "uge"};
      static const std::set<std::string> pf = {"oeq", "ogt", "oge", "olt", "ole", "one", "ord", "ueq", "ugt", "uge", "ult", "ule", "une", "uno"};
      if (!(f ? pf : pi).count(op.predicate)) { diag.fail(op.line, "unknown predicate '" + op.predicate + "'"); return false; }
      sc[op.results.at(0)] = scalar("i1");
    } else if (n == "arith.select") {
      if (!need(3)) return false;
      if (!(ots[0].is_scalar() && ots[0].elem == "i1") || ots[1] != decl || ots[2] != decl) { diag.fail(op.line, "arith.select operand types"); return false; }
      sc[op.results.at(0)] = decl;
    } else if (n == "arith.index_cast" || n == "arith.sitofp" || n == "arith.uitofp" || n == "arith.fptosi" ||
               n == "arith.extf" || n == "arith.truncf" || n == "arith.extsi" || n == "arith.trunci") {
      if (!need(1)) return false;
      if (op.types.size() != 2 || ots[0] != op.types[0] || !op.types[1].is_scalar()) { diag.fail(op.line, "'" + n + "' needs `: from to to` scalar types matching its operand"); return false; }
      sc[op.results.at(0)] = op.types[1];
    } else {
      diag.fail(op.line, "unsupported operation '" + n + "' inside an apply region");
      return false;
    }
    return true;
  }

  // ---- scalar ops inside an apply region (recursively through scf.if) --------------------
  bool check_region_ops(const Block& blk, Scope sc, const Op& apply, int rank, bool linear, bool top,
                        const Type& elem_ty, std::vector<Type>* yielded) {
    for (size_t oi = 0; oi < blk.ops.size(); ++oi) {
      const Op& op = *blk.ops[oi];
      const std::string& n = op.name;
      const bool last = oi + 1 == blk.ops.size();
      if (n == "neptune_ir.yield" || n == "scf.yield") {
        if (!last) { diag.fail(op.line, "'" + n + "' must be the last operation of its block"); return false; }
        if ((n == "neptune_ir.yield") != top) { diag.fail(op.line, "'" + n + "' terminates the wrong kind of region"); return false; }
        std::vector<Type> tys;
        for (auto& v : op.operands) { Type t; if (!lookup(sc, op, v, t)) return false; tys.push_back(t); }
        if (n == "neptune_ir.yield") {
          if (tys.size() != 1) { diag.fail(op.line, "'neptune_ir.yield' op MVP: only single-scalar yield is supported"); return false; }
          if (tys[0] != elem_ty) { diag.fail(op.line, "'neptune_ir.yield' op yield operand type must equal apply result element type"); return false; }
        }
        if (yielded) *yielded = tys;
        return true;
      }
      if (linear) {
        // apply inside a linear_opdef: NeptuneIRVerifier.cpp:44-54 whitelist
        static const std::set<std::string> allowed = {"neptune_ir.access", "arith.addf", "arith.addi", "arith.subf",
                                                      "arith.subi", "arith.mulf", "arith.constant"};
        if (!allowed.count(n)) { diag.fail(op.line, "'" + n + "' op op not allowed inside apply for linear_opdef"); return false; }
      }
      if (n == "neptune_ir.access") {
        Type in;
        if (!lookup(sc, op, op.operands[0], in)) return false;
        if (in.kind != TypeKind::Temp) { diag.fail(op.line, "'neptune_ir.access' op access input must be TempType"); return false; }
        if ((int)op.offsets.size() != rank || in.bounds.rank() != rank) {
          diag.fail(op.line, "'neptune_ir.access' op offsets rank must match apply bounds rank");
          return false;
        }
        for (int64_t off : op.offsets)
          if (off < -(1 << 20) || off > (1 << 20)) { diag.fail(op.line, "'neptune_ir.access' op offset outside the supported range [-2^20, 2^20]"); return false; }
        if (op.types.size() != 2 || op.types[0] != in) { diag.fail(op.line, "'neptune_ir.access' op operand type does not match its declared type"); return false; }
        if (!op.types[1].is_scalar() || op.types[1].elem != in.elem) {
          diag.fail(op.line, "'neptune_ir.access' op result type must equal input Temp element type");
          return false;
        }
        sc[op.results.at(0)] = op.types[1];
        continue;
      }
      if (n.compare(0, 11, "neptune_ir.") == 0) {
        diag.fail(op.line, "'" + n + "' op unexpected NeptuneIR op inside nonlinear apply-like region (only access/yield allowed)");
        return false;
      }
      if (n == "arith.constant") {
        const Type& t = op.types.at(0);
        if (!t.is_scalar()) { diag.fail(op.line, "arith.constant of non-scalar type"); return false; }
        sc[op.results.at(0)] = t;
        continue;
      }
      if (n == "scf.if") {
        Type c;
        if (!lookup(sc, op, op.operands[0], c)) return false;
        if (!(c.is_scalar() && c.elem == "i1")) { diag.fail(op.line, "scf.if condition must be i1"); return false; }
        if (!op.results.empty() && op.regions.size() != 2) { diag.fail(op.line, "scf.if with results needs an else region"); return false; }
        if (op.results.size() != op.types.size()) { diag.fail(op.line, "scf.if result count does not match its result types"); return false; }
        for (auto& r : op.regions) {
          std::vector<Type> y;
          if (!check_region_ops(*r, sc, apply, rank, linear, false, elem_ty, &y)) return false;
          if (y.size() != op.types.size()) { diag.fail(op.line, "scf.yield operand count does not match scf.if results"); return false; }
          for (size_t i = 0; i < y.size(); ++i)
            if (y[i] != op.types[i]) { diag.fail(op.line, "scf.yield operand type does not match scf.if result type"); return false; }
        }
        for (size_t i = 0; i < op.results.size(); ++i) sc[op.results[i]] = op.types[i];
        continue;
      }
      if (!check_arith(op, sc)) return false;
      if (linear && n == "arith.mulf") {
        // VerifyAndAnnotate.cpp:183-186: one factor must be a constant
        auto is_const = [&](const std::string& v) {
          for (auto& o : blk.ops)
            if (!o->results.empty() && o->results[0] == v) return o->name == "arith.constant";
          return false;
        };
        if (!is_const(op.operands[0]) && !is_const(op.operands[1])) {
          diag.fail(op.line, "'arith.mulf' op MulFOp in linear region must multiply by a constant");
          return false;
        }
      }
    }
    diag.fail(apply.line, top ? "'neptune_ir.apply' op apply-like region must terminate with neptune_ir.yield" : "scf.if region must terminate with scf.yield");
    return false;
  }

  bool check_apply(const Op& op, const Scope& outer, bool linear) {
    auto bit = op.attrs.find("bounds");
    if (bit == op.attrs.end() || bit->second.kind != AttrValue::BoundsK) { diag.fail(op.line, "'neptune_ir.apply' op missing required 'bounds' attribute"); return false; }
    const Bounds& b = bit->second.bounds;
    const int rank = b.rank();
    if (rank == 0) { diag.fail(op.line, "'neptune_ir.apply' op 0-D apply not supported"); return false; }
    const size_t nin = op.operands.size();
    if (nin == 0) { diag.fail(op.line, "'neptune_ir.apply' op needs at least one input (copy-through source)"); return false; }
    if (op.types.size() != nin + 1) { diag.fail(op.line, "'neptune_ir.apply' op functional type does not match its operands"); return false; }
    for (size_t k = 0; k < nin; ++k) {
      Type t;
      if (!lookup(outer, op, op.operands[k], t)) return false;
      if (t.kind != TypeKind::Temp) { diag.fail(op.line, "'neptune_ir.apply' op inputs must be temps"); return false; }
      if (t != op.types[k]) { diag.fail(op.line, "'neptune_ir.apply' op operand #" + std::to_string(k) + " type does not match the functional type"); return false; }
      if (t.bounds.rank() != rank) { diag.fail(op.line, "'neptune_ir.apply' op input rank differs from the bounds rank"); return false; }
    }
    const Type& res = op.types[nin];
    if (res.kind != TypeKind::Temp || res.bounds.rank() != rank) { diag.fail(op.line, "'neptune_ir.apply' op result must be a temp of the bounds' rank"); return false; }
    for (int d = 0; d < rank; ++d) {
      // copy-through casts input 0 to the result type (DataflowLowering.cpp:283-287)
      if (res.bounds.ub[d] - res.bounds.lb[d] != op.types[0].bounds.ub[d] - op.types[0].bounds.lb[d]) {
        diag.fail(op.line, "'neptune_ir.apply' op result shape must equal input 0's shape (copy-through)");
        return false;
      }
    }
    if (res.elem != op.types[0].elem) { diag.fail(op.line, "'neptune_ir.apply' op result element type must equal input 0's (copy-through)"); return false; }
    const Block& blk = *op.regions.at(0);
    if (blk.args.size() != (size_t)rank + nin) {
      diag.fail(op.line, "'neptune_ir.apply' op apply-like region block arg count must be (bounds rank + number of inputs) = " +
                             std::to_string(rank + nin) + ", but got " + std::to_string(blk.args.size()));
      return false;
    }
    Scope sc;  // IsolatedFromAbove: the region sees only its own arguments
    for (int d = 0; d < rank; ++d) {
      if (!(blk.args[d].type.is_scalar() && blk.args[d].type.elem == "index")) { diag.fail(op.line, "'neptune_ir.apply' op region arg #" + std::to_string(d) + " must be index"); return false; }
      sc[blk.args[d].name] = blk.args[d].type;
    }
    for (size_t k = 0; k < nin; ++k) {
      if (blk.args[rank + k].type != op.types[k]) {
        diag.fail(op.line, "'neptune_ir.apply' op region input arg #" + std::to_string(rank + k) + " type mismatch: expect " +
                               op.types[k].str() + " but got " + blk.args[rank + k].type.str());
        return false;
      }
      sc[blk.args[rank + k].name] = blk.args[rank + k].type;
    }
    return check_region_ops(blk, sc, op, rank, linear, true, scalar(res.elem), nullptr);
  }

  bool check_function(const Function& f) {
    Scope sc;
    const bool opdef = f.kind != FuncKind::Func;
    const bool linear = f.kind == FuncKind::LinearOpDef;
    if (f.body.args.size() != f.arg_types.size()) {
      diag.fail(f.line, "'neptune_ir." + std::string(linear ? "linear" : "nonlinear") + "_opdef' op block arg count must match function inputs");
      return false;
    }
    for (size_t i = 0; i < f.body.args.size(); ++i) {
      if (f.body.args[i].type != f.arg_types[i]) { diag.fail(f.line, "@" + f.name + ": block argument types must match function inputs"); return false; }
      sc[f.body.args[i].name] = f.arg_types[i];
    }
    bool returned = false;
    for (size_t oi = 0; oi < f.body.ops.size(); ++oi) {
      const Op& op = *f.body.ops[oi];
      const std::string& n = op.name;
      if (op.opaque) {
        if (linear) { diag.fail(op.line, "'" + n + "' op operation not allowed in linear_opdef body"); return false; }
        for (auto& r : op.results) { Type t; sc[r] = t; }  // type unknown to this front end
        continue;
      }
      {
        // values produced by solver / time-stepping ops have no type here: an op that consumes one
        // cannot be checked (the function is not lowered anyway, emit_hip.cpp lowerable())
        bool unknown = false;
        for (auto& v : op.operands) {
          auto it = sc.find(v);
          if (it != sc.end() && it->second.kind == TypeKind::None) unknown = true;
        }
        if (unknown) {
          for (auto& r : op.results) { Type t; sc[r] = t; }
          if (n == "neptune_ir.return" || n == "func.return" || n == "return") returned = true;
          continue;
        }
      }
      if (linear) {
        static const std::set<std::string> allowed = {"neptune_ir.access", "neptune_ir.apply", "neptune_ir.apply_linear",
                                                      "neptune_ir.yield", "neptune_ir.return", "neptune_ir.reduce", "arith.addf", "arith.addi",
                                                      "arith.subf", "arith.subi", "arith.mulf", "arith.constant"};
        if (!allowed.count(n)) { diag.fail(op.line, "'" + n + "' op operation not allowed in linear_opdef body"); return false; }
      }
      if (n == "neptune_ir.as_tensor" || n == "neptune_ir.from_tensor") {
        // temp <-> ranked tensor of the temp's shape and element type (NeptuneIROps.td:540-556, 575-591); like the
        // reference, which keeps both as casts (DataflowLowering.cpp:705-733), they are aliases of the same buffer
        Type in;
        if (!lookup(sc, op, op.operands[0], in)) return false;
        if (in != op.types.at(0)) { diag.fail(op.line, "'" + n + "' op operand type does not match its declared type"); return false; }
        const Type& out = op.types.at(1);
        const Type& tmp = n == "neptune_ir.as_tensor" ? in : out;
        const Type& ten = n == "neptune_ir.as_tensor" ? out : in;
        if (tmp.kind != TypeKind::Temp || ten.kind != TypeKind::MemRef || !ten.tensor) {
          diag.fail(op.line, "'" + n + "' op converts between a temp and a ranked tensor");
          return false;
        }
        if (tmp.elem != ten.elem) { diag.fail(op.line, "'" + n + "' op element type mismatch"); return false; }
        bool same = tmp.rank() == ten.rank();
        for (int d = 0; same && d < tmp.rank(); ++d) same = ten.shape[d] == tmp.bounds.ub[d] - tmp.bounds.lb[d];
        if (!same) { diag.fail(op.line, "'" + n + "' op tensor shape must equal the extents of the temp's bounds"); return false; }
        sc[op.results.at(0)] = out;
      } else if (n == "neptune_ir.wrap" || n == "neptune_ir.unwrap" || n == "neptune_ir.load") {
        Type in;
        if (!lookup(sc, op, op.operands[0], in)) return false;
        if (in != op.types.at(0)) { diag.fail(op.line, "'" + n + "' op operand type does not match its declared type"); return false; }
        const Type& out = op.types.at(1);
        const TypeKind want_in = n == "neptune_ir.wrap" ? TypeKind::MemRef : TypeKind::Field;
        const TypeKind want_out = n == "neptune_ir.wrap" ? TypeKind::Field : (n == "neptune_ir.load" ? TypeKind::Temp : TypeKind::MemRef);
        if (in.kind != want_in || out.kind != want_out) { diag.fail(op.line, "'" + n + "' op has the wrong operand/result kinds"); return false; }
        if (in.elem != out.elem || in.rank() != out.rank()) { diag.fail(op.line, "'" + n + "' op element type / rank mismatch"); return false; }
        sc[op.results.at(0)] = out;
      } else if (n == "neptune_ir.apply") {
        if (!check_apply(op, sc, linear)) return false;
        sc[op.results.at(0)] = op.types.back();
      } else if (n == "neptune_ir.time_advance") {
        // explicit method: result = state + dt * rhs(state)   (HighLevelConvertion.cpp:77-120)
        Type st, dt;
        if (op.operands.size() != 2 || !lookup(sc, op, op.operands[0], st) || !lookup(sc, op, op.operands[1], dt)) {
          diag.fail(op.line, "'neptune_ir.time_advance' op expects a state and a time step");
          return false;
        }
        if (st.kind != TypeKind::Temp || st != op.types.at(0) || op.types.at(2) != st) { diag.fail(op.line, "'neptune_ir.time_advance' op state / result must be the same temp type"); return false; }
        if (!(dt.is_scalar() && dt.elem == "f64") || op.types.at(1) != dt) { diag.fail(op.line, "'neptune_ir.time_advance' op dt must be f64"); return false; }
        if (st.elem != "f64") { diag.fail(op.line, "'neptune_ir.time_advance' op explicit method is f64 only (dt is f64)"); return false; }
        const Function* callee = m.find(op.callee);
        if (!callee || callee->kind == FuncKind::Func) { diag.fail(op.line, "'neptune_ir.time_advance' op rhs must reference linear_opdef or nonlinear_opdef"); return false; }
        if (callee->arg_types.size() != 1 || callee->result_types.size() != 1 || callee->arg_types[0] != st || callee->result_types[0] != st) {
          diag.fail(op.line, "'neptune_ir.time_advance' op rhs @" + op.callee + " must map the state type to itself");
          return false;
        }
        sc[op.results.at(0)] = st;
      } else if (n == "neptune_ir.apply_linear" || n == "neptune_ir.apply_nonlinear") {
        const Function* callee = m.find(op.callee);
        if (!callee) { diag.fail(op.line, "'" + n + "' op unresolved symbol @" + op.callee); return false; }
        if (callee->kind == FuncKind::Func) { diag.fail(op.line, "'" + n + "' op @" + op.callee + " is not an opdef"); return false; }
        if (callee->arg_types.size() != op.operands.size() || op.types.size() != op.operands.size() + callee->result_types.size() ||
            op.results.size() != callee->result_types.size()) {
          diag.fail(op.line, "'" + n + "' op signature does not match @" + op.callee);
          return false;
        }
        for (size_t k = 0; k < op.operands.size(); ++k) {
          Type t;
          if (!lookup(sc, op, op.operands[k], t)) return false;
          if (t != callee->arg_types[k] || op.types[k] != t) { diag.fail(op.line, "'" + n + "' op argument #" + std::to_string(k) + " type does not match @" + op.callee); return false; }
        }
        for (size_t r = 0; r < op.results.size(); ++r) {
          if (op.types[op.operands.size() + r] != callee->result_types[r]) { diag.fail(op.line, "'" + n + "' op result type does not match @" + op.callee); return false; }
          sc[op.results[r]] = callee->result_types[r];
        }
      } else if (n == "neptune_ir.store") {
        Type v, fld;
        if (!lookup(sc, op, op.operands[0], v) || !lookup(sc, op, op.operands[1], fld)) return false;
        if (v.kind != TypeKind::Temp || fld.kind != TypeKind::Field) { diag.fail(op.line, "'neptune_ir.store' op store expects TempType -> FieldType"); return false; }
        if (v != op.types.at(0) || fld != op.types.at(1)) { diag.fail(op.line, "'neptune_ir.store' op operand types do not match the declared types"); return false; }
        if (v.elem != fld.elem || v.rank() != fld.rank()) { diag.fail(op.line, "'neptune_ir.store' op element type / rank mismatch"); return false; }
        auto bit = op.attrs.find("bounds");
        if (bit != op.attrs.end()) {
          if (bit->second.kind != AttrValue::BoundsK || bit->second.bounds.rank() != v.rank()) { diag.fail(op.line, "'neptune_ir.store' op bounds rank mismatch"); return false; }
        } else {
          for (int d = 0; d < v.rank(); ++d)
            if (v.bounds.ub[d] - v.bounds.lb[d] != fld.bounds.ub[d] - fld.bounds.lb[d]) { diag.fail(op.line, "'neptune_ir.store' op whole-buffer store needs equal shapes"); return false; }
        }
      } else if (n == "arith.constant") {
        sc[op.results.at(0)] = op.types.at(0);
      } else if (n == "neptune_ir.reduce") {
        Type in;
        if (!lookup(sc, op, op.operands[0], in)) return false;
        if (in.kind != TypeKind::Temp || in != op.types.at(0)) { diag.fail(op.line, "'neptune_ir.reduce' op reduce input must be TempType"); return false; }
        auto kit = op.attrs.find("kind");
        if (kit == op.attrs.end() || kit->second.kind != AttrValue::String || kit->second.s != "sum") {
          diag.fail(op.line, "'neptune_ir.reduce' op MVP reduce only supports kind=\"sum\"");
          return false;
        }
        auto bit = op.attrs.find("bounds");
        if (bit != op.attrs.end() && (bit->second.kind != AttrValue::BoundsK || bit->second.bounds.rank() != in.rank())) {
          diag.fail(op.line, "'neptune_ir.reduce' op bounds rank mismatch in reduce");
          return false;
        }
        if (!op.types.at(1).is_scalar() || op.types[1].elem != in.elem) { diag.fail(op.line, "'neptune_ir.reduce' op result type must equal the input's element type"); return false; }
        sc[op.results.at(0)] = op.types[1];
      } else if ((n.compare(0, 6, "arith.") == 0 || n.compare(0, 5, "math.") == 0) && op.regions.empty()) {
        if (!check_arith(op, sc)) return false;   // scalar arithmetic on reduce results / constants
      } else if (n == "neptune_ir.return" || n == "func.return" || n == "return") {
        if (oi + 1 != f.body.ops.size()) { diag.fail(op.line, "return must be the last operation of @" + f.name); return false; }
        if (opdef && n != "neptune_ir.return") { diag.fail(op.line, "@" + f.name + ": body must terminate with neptune_ir.return"); return false; }
        if (op.operands.size() != f.result_types.size()) { diag.fail(op.line, "@" + f.name + ": return operand count must match function results"); return false; }
        for (size_t r = 0; r < op.operands.size(); ++r) {
          Type t;
          if (!lookup(sc, op, op.operands[r], t)) return false;
          if (t != f.result_types[r]) { diag.fail(op.line, "@" + f.name + ": return operand types must match function results"); return false; }
        }
        returned = true;
      } else {
        diag.fail(op.line, "unsupported operation '" + n + "' at function level");
        return false;
      }
    }
    if (!returned) { diag.fail(f.line, "@" + f.name + (opdef ? ": body must terminate with neptune_ir.return" : ": missing return")); return false; }
    return true;
  }
};

}  // namespace

bool verify_module(const Module& m, Diag& diag) {
  Verifier v(m, diag);
  for (auto& f : m.funcs)
    if (!v.check_function(*f)) return false;
  return diag.ok;
}

}  // namespace neptune_lowering

// emit_hip.cpp -- the `backend = hip` dataflow lowering: NeptuneIR module -> one HIP translation
// unit whose host side calls the hand-written stencil kernels.
//
// It takes the place of the reference's NeptuneIRDataflowLoweringPass
// (lib/Passes/DataflowLowering.cpp:739-819, the `backend == gpu` slot at :803-804 that was never
// wired, lib/Pipeline/NeptuneIRPassesPipeline.cpp:22-26) together with the part of
// StructureLowering (lib/Passes/StructureLowering.cpp:30-124) it depends on:
//
//   linear_opdef / nonlinear_opdef @A      -> exported extern "C" symbol A + internal A__impl
//   apply_linear / apply_nonlinear @A(x)   -> call of A__impl (device-resident, no staging)
//   wrap / unwrap / load                   -> aliases                      (:131-159)
//   apply                                  -> Body functor (one C++ statement per region op, in
//                                             textual order) + run_apply<Body,...>: ONE kernel that
//                                             also does the copy-through of input 0 (:258-448)
//   store                                  -> elided when the producing apply can write straight
//                                             into the destination field, else a device copy (:165-220)
//   reduce {kind = "sum"}                  -> fixed-tree device sum (:589-698); of a single-use apply
//                                             result: ONE kernel that evaluates the body and sums
//   time_advance {method = 0, rhs = @A}    -> u + dt*A(u): one kernel when @A is a single apply of
//                                             the state, else call + axpy apply
//                                             (lib/Passes/HighLevelConvertion.cpp:77-120)
//   func.func @entry                       -> exported symbol with expanded memref arguments and a
//                                             memref struct result (upstream func-to-llvm ABI,
//                                             NeptuneIRPassesPipeline.cpp:36-40)
//
// The arithmetic inside a Body is emitted as written: no reassociation, no contraction (the TU is
// compiled with -ffp-contract=off), constants as hexadecimal floating literals.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <set>
#include <algorithm>
#include <sstream>

#include "lowering.h"

namespace neptune_lowering {
namespace {

std::string cname(const std::string& ssa) {  // %lap_i -> v_lap_i
  std::string s = "v_";
  for (size_t i = 1; i < ssa.size(); ++i) s += (std::isalnum((unsigned char)ssa[i]) ? ssa[i] : '_');
  return s;
}
std::string ctype(const std::string& elem) {
  if (elem == "f64") return "double";
  if (elem == "f32") return "float";
  if (elem == "index" || elem == "i64") return "int64_t";
  if (elem == "i32") return "int32_t";
  if (elem == "i1") return "bool";
  return "void";
}
int esize(const std::string& elem) { return elem == "f32" ? 4 : 8; }
std::string dtype_macro(const std::string& elem) { return elem == "f32" ? "NEPTUNE_HIP_F32" : "NEPTUNE_HIP_F64"; }

std::string float_literal(const std::string& lit, const std::string& elem, bool& ok) {
  ok = true;
  if (lit.size() > 2 && (lit.compare(0, 2, "0x") == 0 || lit.compare(0, 3, "-0x") == 0)) { ok = false; return ""; }
  char buf[64];
  const double d = std::strtod(lit.c_str(), nullptr);
  if (elem == "f32") {
    const float f = (float)d;  // MLIR parses the literal as a double and rounds it to f32
    if (std::isinf(f) || std::isnan(f)) { ok = false; return ""; }
    snprintf(buf, sizeof(buf), "%af", (double)f);
  } else {
    if (std::isinf(d) || std::isnan(d)) { ok = false; return ""; }
    snprintf(buf, sizeof(buf), "%a", d);
  }
  return buf;
}

std::string box_init(const Bounds& b) {
  std::ostringstream o;
  o << "{" << b.rank() << ", {";
  for (int d = 0; d < 6; ++d) o << (d ? ", " : "") << (d < b.rank() ? b.lb[d] : 0);
  o << "}, {";
  for (int d = 0; d < 6; ++d) o << (d ? ", " : "") << (d < b.rank() ? b.ub[d] : 1);
  o << "}}";
  return o.str();
}

struct Footprint {
  int nin = 0, rank = 0;
  int radius[4][3];      // all accesses
  int top_radius[4][3];  // unconditional accesses only; -1 = none
  int top_lo[4][3], top_hi[4][3];  // ... as offsets: most negative / most positive; hi < lo = none
  bool box = false;
  int halo_input = -1;
  unsigned halo_mask = 0;
  int halo_inputs = 0;
  int R[3] = {0, 0, 0};  // shared radii on the kernel's (I,J,K) axes
  bool march_ok = true;
  int lead = 0;          // leading (batch / component) dimensions of an apply of rank 4..6: peeled off on the host
  // rank 4..6 with offsets along a leading dimension: a stencil in more than three dimensions, run by the rank-generic
  // kernel (kernels/apply_nd.hpp) -- what its unconditional accesses reach per input and dimension, its reach along dim 0
  bool nd = false;
  int nd_lo[4][6], nd_hi[4][6];
  int nd_halo0 = 0;
  bool exact = true;     // no elementary functions in the body
};

struct Emitter {
  const Module& m;
  Diag& diag;
  LowerInfo& info;
  std::ostringstream bodies, funcs, geom_entries;
  bool saw_elementary = false;  // set by emit_op when it emits exp/log/sin/cos/tanh/powf
  bool nd_body = false;         // emitting the body of a Footprint::nd apply: accesses carry one offset per dimension
  int box_counter = 0;
  std::ostringstream consts;
  Emitter(const Module& mm, Diag& d, LowerInfo& i) : m(mm), diag(d), info(i) {}

  std::string new_box(const Bounds& b) {
    std::string name = "kBox" + std::to_string(box_counter++);
    consts << "static const nl::Box " << name << " = " << box_init(b) << ";\n";
    return name;
  }

  // ---- apply region -> Body functor ----------------------------------------------------
  void scan_accesses(const Block& blk, const std::map<std::string, int>& temp_index, Footprint& fp, bool top) {
    for (auto& op : blk.ops) {
      if (op->name == "neptune_ir.access") {
        const int k = temp_index.at(op->operands[0]);
        int nz = 0;
        for (int d = 0; d < fp.rank; ++d) {
          const int a = (int)std::llabs(op->offsets[fp.lead + d]);
          if (a) ++nz;
          if (a > fp.radius[k][d]) fp.radius[k][d] = a;
          if (top && a > fp.top_radius[k][d]) fp.top_radius[k][d] = a;  // starts at -1: offset 0 counts
          if (top) {
            const int off = (int)op->offsets[fp.lead + d];
            if (fp.top_hi[k][d] < fp.top_lo[k][d]) fp.top_lo[k][d] = fp.top_hi[k][d] = off;   // the first unconditional access
            else { fp.top_lo[k][d] = std::min(fp.top_lo[k][d], off); fp.top_hi[k][d] = std::max(fp.top_hi[k][d], off); }
          }
        }
        if (nz > 1) fp.box = true;
      }
      for (auto& r : op->regions) scan_accesses(*r, temp_index, fp, false);
    }
  }

  bool emit_region_ops(const Block& blk, std::ostringstream& o, const std::string& ind,
                       const std::map<std::string, int>& temp_index, const std::map<std::string, int>& index_arg,
                       const std::vector<std::string>* if_results) {
    for (auto& opp : blk.ops)
      if (!emit_op(*opp, o, ind, temp_index, index_arg, if_results)) return false;
    return true;
  }

  // one scalar op -> one C++ statement (also used for scalar arithmetic at function level, where
  // the same text is valid host code)
  bool emit_op(const Op& op, std::ostringstream& o, const std::string& ind, const std::map<std::string, int>& temp_index,
               const std::map<std::string, int>& index_arg, const std::vector<std::string>* if_results) {
    {
      const std::string& n = op.name;
      auto val = [&](const std::string& v) -> std::string {
        auto it = index_arg.find(v);
        if (it != index_arg.end())
          return it->second < 0 ? "lead[" + std::to_string(-it->second - 1) + "]" : "a.template idx<" + std::to_string(it->second) + ">()";
        return cname(v);
      };
      auto res = [&]() { return cname(op.results.at(0)); };
      if (n == "neptune_ir.access") {
        o << ind << "const " << ctype(op.types[1].elem) << " " << res() << " = a.template get<" << temp_index.at(op.operands[0]);
        for (size_t d = (op.offsets.size() > 3 && !nd_body) ? op.offsets.size() - 3 : 0; d < op.offsets.size(); ++d) o << ", " << op.offsets[d];
        o << ">();\n";
      } else if (n == "arith.constant") {
        const Type& t = op.types[0];
        if (t.elem == "f64" || t.elem == "f32") {
          bool ok;
          std::string lit = float_literal(op.literal, t.elem, ok);
          if (!ok) { diag.fail(op.line, "unsupported floating-point constant '" + op.literal + "'"); return false; }
          o << ind << "const " << ctype(t.elem) << " " << res() << " = " << lit << ";  // " << op.literal << "\n";
        } else if (t.elem == "i1") {
          o << ind << "const bool " << res() << " = " << ((op.literal == "true" || op.literal == "1") ? "true" : "false") << ";\n";
        } else {
          o << ind << "const " << ctype(t.elem) << " " << res() << " = (" << ctype(t.elem) << ")" << op.literal << "LL;\n";
        }
      } else if (n == "arith.addf" || n == "arith.subf" || n == "arith.mulf" || n == "arith.divf" || n == "arith.addi" ||
                 n == "arith.subi" || n == "arith.muli") {
        const char* sym = (n.find("add") != std::string::npos) ? "+" : (n.find("sub") != std::string::npos) ? "-"
                          : (n.find("mul") != std::string::npos) ? "*" : "/";
        o << ind << "const " << ctype(op.types[0].elem) << " " << res() << " = " << val(op.operands[0]) << " " << sym << " "
          << val(op.operands[1]) << ";\n";
      } else if (n == "arith.andi" || n == "arith.ori" || n == "arith.xori") {
        const bool b1 = op.types[0].elem == "i1";
        const char* sym = n == "arith.andi" ? (b1 ? "&&" : "&") : n == "arith.ori" ? (b1 ? "||" : "|") : (b1 ? "!=" : "^");
        o << ind << "const " << ctype(op.types[0].elem) << " " << res() << " = " << val(op.operands[0]) << " " << sym << " "
          << val(op.operands[1]) << ";\n";
      } else if (n == "arith.negf") {
        o << ind << "const " << ctype(op.types[0].elem) << " " << res() << " = -" << val(op.operands[0]) << ";\n";
      } else if (n == "arith.maximumf" || n == "arith.minimumf" || n == "arith.maxnumf" || n == "arith.minnumf") {
        o << ind << "const " << ctype(op.types[0].elem) << " " << res() << " = neptune_hip::ops::" << n.substr(6) << "("
          << val(op.operands[0]) << ", " << val(op.operands[1]) << ");\n";
      } else if (n == "math.copysign") {
        o << ind << "const " << ctype(op.types[0].elem) << " " << res() << " = neptune_hip::ops::copysign(" << val(op.operands[0])
          << ", " << val(op.operands[1]) << ");\n";
      } else if (n == "math.powf") {
        saw_elementary = true;
        o << ind << "const " << ctype(op.types[0].elem) << " " << res() << " = neptune_hip::ops::powf(" << val(op.operands[0]) << ", "
          << val(op.operands[1]) << ");\n";
      } else if (n == "math.sqrt" || n == "math.absf" || n == "math.floor" || n == "math.ceil" || n == "math.exp" || n == "math.log" || n == "math.sin" || n == "math.cos" ||
                 n == "math.tanh") {
        if (n != "math.sqrt" && n != "math.absf" && n != "math.floor" && n != "math.ceil") saw_elementary = true;
        o << ind << "const " << ctype(op.types[0].elem) << " " << res() << " = neptune_hip::ops::" << n.substr(5) << "("
          << val(op.operands[0]) << ");\n";
      } else if (n == "arith.cmpf") {
        const std::string a = val(op.operands[0]), b = val(op.operands[1]), &p = op.predicate;
        std::string e;
        if (p == "oeq") e = a + " == " + b; else if (p == "ogt") e = a + " > " + b; else if (p == "oge") e = a + " >= " + b;
        else if (p == "olt") e = a + " < " + b; else if (p == "ole") e = a + " <= " + b;
        else if (p == "one") e = "(" + a + " < " + b + " || " + a + " > " + b + ")";
        else if (p == "ord") e = "(" + a + " == " + a + " && " + b + " == " + b + ")";
        else if (p == "ueq") e = "!(" + a + " < " + b + " || " + a + " > " + b + ")";
        else if (p == "ugt") e = "!(" + a + " <= " + b + ")"; else if (p == "uge") e = "!(" + a + " < " + b + ")";
        else if (p == "ult") e = "!(" + a + " >= " + b + ")"; else if (p == "ule") e = "!(" + a + " > " + b + ")";
        else if (p == "une") e = a + " != " + b; else e = "(" + a + " != " + a + " || " + b + " != " + b + ")";
        o << ind << "const bool " << res() << " = " << e << ";\n";
      } else if (n == "arith.cmpi") {
        const std::string &p = op.predicate;
        const bool uns = p[0] == 'u';
        const std::string ct = ctype(op.types[0].elem);
        const std::string cast = uns ? "(uint64_t)" : "";
        std::string sym = (p == "eq") ? "==" : (p == "ne") ? "!=" : (p.substr(1) == "lt") ? "<" : (p.substr(1) == "le") ? "<="
                          : (p.substr(1) == "gt") ? ">" : ">=";
        o << ind << "const bool " << res() << " = " << cast << val(op.operands[0]) << " " << sym << " " << cast
          << val(op.operands[1]) << ";\n";
        (void)ct;
      } else if (n == "arith.select") {
        o << ind << "const " << ctype(op.types[0].elem) << " " << res() << " = " << val(op.operands[0]) << " ? "
          << val(op.operands[1]) << " : " << val(op.operands[2]) << ";\n";
      } else if (n == "arith.index_cast" || n == "arith.sitofp" || n == "arith.fptosi" || n == "arith.extf" ||
                 n == "arith.truncf" || n == "arith.extsi" || n == "arith.trunci") {
        o << ind << "const " << ctype(op.types[1].elem) << " " << res() << " = (" << ctype(op.types[1].elem) << ")"
          << val(op.operands[0]) << ";\n";
      } else if (n == "arith.uitofp") {
        o << ind << "const " << ctype(op.types[1].elem) << " " << res() << " = (" << ctype(op.types[1].elem) << ")(uint64_t)"
          << val(op.operands[0]) << ";\n";
      } else if (n == "scf.if") {
        for (size_t r = 0; r < op.results.size(); ++r) o << ind << ctype(op.types[r].elem) << " " << cname(op.results[r]) << ";\n";
        o << ind << "if (" << val(op.operands[0]) << ") {\n";
        if (!emit_region_ops(*op.regions[0], o, ind + "  ", temp_index, index_arg, &op.results)) return false;
        o << ind << "}";
        if (op.regions.size() > 1) {
          o << " else {\n";
          if (!emit_region_ops(*op.regions[1], o, ind + "  ", temp_index, index_arg, &op.results)) return false;
          o << ind << "}";
        }
        o << "\n";
      } else if (n == "scf.yield") {
        for (size_t r = 0; r < op.operands.size() && if_results; ++r)
          o << ind << cname((*if_results)[r]) << " = " << val(op.operands[r]) << ";\n";
      } else if (n == "neptune_ir.yield") {
        o << ind << "return " << val(op.operands[0]) << ";\n";
      } else {
        diag.fail(op.line, "cannot emit '" + n + "'");
        return false;
      }
    }
    return true;
  }

  // checks + footprint of one apply (no emission): which inputs need a ring, the shared radii on the
  // kernel's axes, whether the march kernel can take it
  bool analyze_apply(const Op& apply, Footprint& fp, std::map<std::string, int>& temp_index,
                     std::map<std::string, int>& index_arg) {
    const Block& blk = *apply.regions[0];
    const Bounds& b = apply.attrs.at("bounds").bounds;
    const int rank = b.rank();
    const int nin = (int)apply.operands.size();
    const Type& res = apply.types[nin];
    if (rank > 6) { diag.fail(apply.line, "apply of rank " + std::to_string(rank) + " (the HIP backend supports rank 1..6)"); return false; }
    if (nin > 4) { diag.fail(apply.line, "apply with more than 4 inputs"); return false; }
    if (res.elem != "f64" && res.elem != "f32") { diag.fail(apply.line, "apply element type " + res.elem + " (f64 and f32 are supported)"); return false; }
    for (int k = 0; k < nin; ++k)
      if (apply.types[k].elem != res.elem) { diag.fail(apply.line, "apply inputs of mixed element types"); return false; }
    fp = Footprint();
    fp.nin = nin;
    // rank 4..6: the leading rank-3 dimensions are batch / component dimensions -- no access may have an offset along them
    // (the kernels are rank 1..3); the host launches one rank-3 apply per leading index (run_apply_batched) and the body
    // sees the leading indices as members.  Index argument d maps to -(d+1) for a leading dimension.
    const int lead = rank > 3 ? rank - 3 : 0;
    fp.lead = lead;
    fp.rank = rank - lead;
    for (int k = 0; k < 4; ++k)
      for (int d = 0; d < 3; ++d) { fp.radius[k][d] = 0; fp.top_radius[k][d] = -1; fp.top_lo[k][d] = 1; fp.top_hi[k][d] = -1; }
    for (int d = 0; d < rank; ++d) index_arg[blk.args[d].name] = d < lead ? -(d + 1) : d - lead;
    for (int k = 0; k < nin; ++k) temp_index[blk.args[rank + k].name] = k;
    if (lead > 0) {
      // an offset along a leading dimension makes it a stencil in more than three dimensions: the rank-generic kernel
      for (int k = 0; k < 4; ++k)
        for (int d = 0; d < 6; ++d) { fp.nd_lo[k][d] = 1; fp.nd_hi[k][d] = -1; }
      std::function<void(const Block&, bool)> scan_nd = [&](const Block& b, bool top) {
        for (auto& op : b.ops) {
          if (op->name == "neptune_ir.access") {
            const int k = temp_index.at(op->operands[0]);
            for (int d = 0; d < rank; ++d) {
              const int off = (int)op->offsets[d];
              if (d < lead && off != 0) fp.nd = true;
              if (d == 0) fp.nd_halo0 = std::max(fp.nd_halo0, std::abs(off));
              if (top) {
                if (fp.nd_hi[k][d] < fp.nd_lo[k][d]) fp.nd_lo[k][d] = fp.nd_hi[k][d] = off;
                else { fp.nd_lo[k][d] = std::min(fp.nd_lo[k][d], off); fp.nd_hi[k][d] = std::max(fp.nd_hi[k][d], off); }
              }
            }
          }
          for (auto& r : op->regions) scan_nd(*r, false);
        }
      };
      scan_nd(blk, true);
      if (fp.nd)
        for (int d = 0; d < rank; ++d) index_arg[blk.args[d].name] = d;   // every index argument through the accessor
    }
    // inputs never read unconditionally keep top_radius -1 ("not accessed": nothing to check)
    scan_accesses(blk, temp_index, fp, true);
    // rank mapping onto the kernel's (I,J,K) axes, see apply_common.hpp AxisMap.
    // every input read at a non-zero offset gets a register ring in the march kernel; the rings share
    // the largest radii
    const int krank = fp.rank;   // the kernel's rank (the last three dimensions of a wider apply)
    int* R = fp.R;
    for (int k = 0; k < nin; ++k) {
      bool any = false;
      for (int d = 0; d < krank; ++d) any = any || fp.radius[k][d] > 0;
      if (!any) continue;
      ++fp.halo_inputs;
      fp.halo_mask |= 1u << k;
      if (fp.halo_input < 0) fp.halo_input = k;
      const int* r = fp.radius[k];
      int m[3] = {0, 0, 0};
      if (fp.rank == 3) { m[0] = r[0]; m[1] = r[1]; m[2] = r[2]; }
      else if (fp.rank == 2) { m[0] = r[0]; m[2] = r[1]; }
      else { m[2] = r[0]; }
      for (int a = 0; a < 3; ++a) R[a] = std::max(R[a], m[a]);
    }
    const int vk = 16 / esize(res.elem);
    // the march kernel keeps 2*R0+1 planes of RJ+2*R1 rows per halo input in registers.  3-D: radius 1 for box
    // stencils, up to 4 for stars (4th/6th/8th-order 13-, 19- and 25-point operators; the 2-row late-J-halo tile
    // holds a radius-3 ring in 209 VGPRs without scratch, a radius-4 ring in 229 with one plane in flight), at most
    // two halo inputs (one beyond radius 1).
    // 1-D / 2-D, where a plane is one row: stars up to radius 4 (8th-order operators; K neighbours beyond one lane
    // vector come through a second wave shift, so the K radius may reach two vectors), up to four halo inputs
    // (one beyond radius 2: the register rings of two wide inputs leave one wave per SIMD), boxes up to radius 2
    // (5x5 windows, one halo input).  Wider footprints use the direct kernel.
    const int rmax = fp.box ? (krank == 3 ? 1 : 2) : 4;
    const int rbig = std::max(R[0], std::max(R[1], R[2]));
    const int hmax = krank == 3 ? (rbig > 1 ? 1 : 2) : ((rbig > 2 || (fp.box && rbig > 1)) ? 1 : 4);
    fp.march_ok = fp.halo_inputs <= hmax && R[0] <= rmax && R[1] <= rmax && R[2] <= (krank == 3 ? rmax : 2 * vk) && R[2] <= 2 * vk;
    // 3-D stars of one halo input beyond that, up to radius 8 (10th- to 16th-order operators): the plane-in-LDS kernel
    // (apply_plane.hpp) keeps only the ring of own cells in registers and reads J / K neighbours from the centre plane in LDS
    if (!fp.march_ok && krank == 3 && !fp.box && fp.halo_inputs == 1 && rbig <= 8) fp.march_ok = true;
    // ... and several inputs read at offsets, a ring and an LDS window each, while two rows per lane of all rings fit the
    // registers: inputs * (2*max(R0,1)+2) <= 21 (apply_plane.hpp plane_capable)
    if (!fp.march_ok && krank == 3 && !fp.box && fp.halo_inputs >= 2 && rbig <= 8 && (R[1] > 0 || R[2] > 0) &&
        fp.halo_inputs * (2 * std::max(R[0], 1) + 2) <= 21)
      fp.march_ok = true;
    // 2-D footprints beyond that, up to radius 8 (stars and boxes, up to four inputs read at offsets): the window of a tile in
    // LDS (neptune_apply_tile2)
    if (!fp.march_ok && krank == 2 && fp.halo_inputs >= 1 && fp.halo_inputs <= 4 && rbig <= 8) fp.march_ok = true;
    // 3-D boxes of one halo input up to radius 2 (125 points): every live plane in LDS (neptune_apply_planes)
    if (!fp.march_ok && krank == 3 && fp.box && fp.halo_inputs == 1 && rbig <= 2 && R[0] >= 1) fp.march_ok = true;
    if (!fp.march_ok) { fp.halo_input = -1; fp.halo_mask = 0; R[0] = R[1] = R[2] = 0; }
    return true;
  }

  // reach of a fused explicit time step: the rhs body's unconditional accesses of the state, and the axpy's read of the centre
  static std::string fused_reach(const Footprint& cfp, int rank) {
    std::ostringstream o;
    o << "{";
    for (int side = 0; side < 2; ++side) {
      o << (side ? ", {" : "{");
      for (int k = 0; k < 4; ++k) {
        o << (k ? ", {" : "{");
        for (int d = 0; d < 3; ++d) {
          int v = side ? -1 : 1;   // not accessed
          if (k == 0 && d < rank) {
            const bool any = cfp.top_hi[0][d] >= cfp.top_lo[0][d];
            v = side ? std::max(any ? cfp.top_hi[0][d] : 0, 0) : std::min(any ? cfp.top_lo[0][d] : 0, 0);
          }
          o << (d ? ", " : "") << v;
        }
        o << "}";
      }
      o << "}";
    }
    o << "}";
    return o.str();
  }

  // reach of an apply along dim 0 (the slab axis), over all of its accesses
  static int halo0_of(const Footprint& fp) {
    int h = 0;
    for (int k = 0; k < fp.nin; ++k) h = std::max(h, fp.radius[k][0]);
    return h;
  }

  bool emit_body(const Op& apply, const std::string& tag, Footprint& fp) {
    const Block& blk = *apply.regions[0];
    const int nin = (int)apply.operands.size();
    const int rank = apply.attrs.at("bounds").bounds.rank();
    const Type& res = apply.types[nin];
    std::map<std::string, int> temp_index, index_arg;
    if (!analyze_apply(apply, fp, temp_index, index_arg)) return false;
    const int* R = fp.R;
    const int halo_inputs = fp.halo_inputs;
    const unsigned halo_mask = fp.halo_mask;

    std::ostringstream& o = bodies;
    o << "// " << tag << ": region of the neptune_ir.apply at line " << apply.line << "\n";
    o << "struct Body_" << tag << " {\n";
    if (rank > 3 && !fp.nd) o << "  int64_t lead[3] = {0, 0, 0};   // indices along the leading (batch) dimensions, set per launch\n";
    o << "  template <class A>\n  __device__ __forceinline__ " << ctype(res.elem) << " operator()(const A& a) const {\n";
    saw_elementary = false;
    nd_body = fp.nd;
    const bool body_ok = emit_region_ops(blk, o, "    ", temp_index, index_arg, nullptr);
    nd_body = false;
    if (!body_ok) return false;
    fp.exact = !saw_elementary;
    o << "  }\n};\n";
    o << "using FP_" << tag << " = neptune_hip::Footprint<" << fp.halo_input << ", " << R[0] << ", " << R[1] << ", " << R[2] << ", "
      << ((fp.box && fp.march_ok) ? "true" : "false") << ", " << (fp.march_ok ? "true" : "false");
    if (halo_inputs > 1 && fp.march_ok) o << ", 0x" << std::hex << halo_mask << std::dec << "u";
    o << ">;\n";
    // what the unconditional accesses reach, per input and dimension: {most negative offsets}, {most positive} (hi < lo: none)
    o << "static const neptune_hip::Reach kTopRadius_" << tag << " = {";
    for (int side = 0; side < 2; ++side) {
      o << (side ? ", {" : "{");
      for (int k = 0; k < 4; ++k) {
        o << (k ? ", {" : "{");
        for (int d = 0; d < 3; ++d)
          o << (d ? ", " : "") << (k < nin && d < fp.rank ? (side ? fp.top_hi[k][d] : fp.top_lo[k][d]) : (side ? -1 : 1));
        o << "}";
      }
      o << "}";
    }
    o << "};\n";
    if (fp.nd) {
      o << "static const neptune_hip::ReachN kNdReach_" << tag << " = {";
      for (int side = 0; side < 2; ++side) {
        o << (side ? ", {" : "{");
        for (int k = 0; k < 4; ++k) {
          o << (k ? ", {" : "{");
          for (int d = 0; d < 6; ++d)
            o << (d ? ", " : "") << (k < nin && d < rank ? (side ? fp.nd_hi[k][d] : fp.nd_lo[k][d]) : (side ? -1 : 1));
          o << "}";
        }
        o << "}";
      }
      o << "};\n";
    }
    o << "\n";
    return true;
  }

  // ---- functions -----------------------------------------------------------------------
  struct ValueInfo {
    Type type;
    int root_arg = -1;  // >= 0: aliases function argument #root_arg
    int def_index = -1;
    int uses = 0;
  };

  bool lowerable(const Function& f, std::string& why, int depth = 0) {
    if (depth > 32) { why = "recursive opdef calls"; return false; }
    for (auto& op : f.body.ops)
      if (op->opaque) { why = "contains '" + op->name + "' (solver / time-stepping op: stays on the host path)"; return false; }
    if (f.result_types.size() > 1) { why = "more than one result"; return false; }
    for (auto& t : f.arg_types)
      if (!(t.kind == TypeKind::MemRef || t.is_tempish())) { why = "argument of type " + t.str() + " (only memref / temp / field arguments are lowered)"; return false; }
    for (auto& t : f.result_types)
      if (!(t.kind == TypeKind::MemRef || t.is_tempish() || (t.is_scalar() && (t.elem == "f64" || t.elem == "f32")))) {
        why = "result of type " + t.str();
        return false;
      }
    for (auto& t : f.arg_types) {
      if (t.rank() < 1 || t.rank() > 6) { why = "rank " + std::to_string(t.rank()) + " argument"; return false; }
      if (t.elem != "f64" && t.elem != "f32") { why = "element type " + t.elem; return false; }
    }
    // every callee must be lowerable too
    for (auto& op : f.body.ops)
      if (!op->callee.empty()) {
        const Function* c = m.find(op->callee);
        std::string w2;
        if (!c || !lowerable(*c, w2, depth + 1)) { why = "calls @" + op->callee + " which is not lowered"; return false; }
      }
    return true;
  }

  struct FusedReduce {
    std::string tag, elem, result_box, bounds_box;
    int rank = 0, nin = 0, halo0 = 0;
  };

  bool emit_function(const Function& f) {
    std::map<std::string, ValueInfo> vals;
    std::map<std::string, FusedReduce> fused_reduce;  // apply result -> how the consuming reduce evaluates it
    std::map<std::string, int> scalar_kind;           // function-level scalars: 0 uniform, 1 bare reduce result, 2 derived from one
    int returned_scalar_kind = -1;
    const int nargs = (int)f.arg_types.size();
    for (int i = 0; i < nargs; ++i) {
      ValueInfo vi;
      vi.type = f.arg_types[i];
      vi.root_arg = i;
      vals[f.body.args[i].name] = vi;
    }
    // use counts + definitions
    for (size_t oi = 0; oi < f.body.ops.size(); ++oi)
      for (auto& v : f.body.ops[oi]->operands) vals[v].uses++;
    // which producer may write straight into which destination
    std::map<int, std::string> dest_of;  // producer op index -> C expression of the destination Val*
    int returned_producer = -1;
    std::map<std::string, int> def_at;
    for (size_t oi = 0; oi < f.body.ops.size(); ++oi)
      for (auto& r : f.body.ops[oi]->results) def_at[r] = (int)oi;
    for (size_t oi = 0; oi < f.body.ops.size(); ++oi) {
      const Op& op = *f.body.ops[oi];
      auto producer_of = [&](const std::string& v) -> int {
        auto it = def_at.find(v);
        if (it == def_at.end()) return -1;
        const Op& p = *f.body.ops[it->second];
        if (p.name == "neptune_ir.apply" || !p.callee.empty()) return it->second;
        return -1;
      };
      if (op.name == "neptune_ir.store" && !op.attrs.count("bounds")) {
        const int p = producer_of(op.operands[0]);
        const std::string& field = op.operands[1];
        const bool field_before = !def_at.count(field) || def_at[field] < p;  // function args precede everything
        if (p >= 0 && vals[op.operands[0]].uses == 1 && field_before) {
          bool clean = true;  // nothing between producer and store may observe the field
          for (int j = p + 1; j < (int)oi; ++j) {
            const std::string& nm = f.body.ops[j]->name;
            if (!(nm == "neptune_ir.wrap" || nm == "neptune_ir.unwrap" || nm == "neptune_ir.load" || nm == "neptune_ir.as_tensor" ||
                  nm == "neptune_ir.from_tensor" || nm == "arith.constant")) clean = false;
          }
          if (clean) dest_of[p] = "&" + cname(field);
        }
      }
      if ((op.name == "neptune_ir.return" || op.name == "func.return" || op.name == "return") && op.operands.size() == 1) {
        // the caller's destination may only be handed down to a producer that is the LAST thing the function
        // computes: an op between it and the return could still read an argument that aliases that destination
        // (the reference gives every apply a private result, DataflowLowering.cpp:281)
        const int p = producer_of(op.operands[0]);
        bool clean = p >= 0 && vals[op.operands[0]].uses == 1;
        for (int j = p + 1; clean && j < (int)oi; ++j) {
          const std::string& nm = f.body.ops[j]->name;
          if (!(nm == "neptune_ir.wrap" || nm == "neptune_ir.unwrap" || nm == "neptune_ir.load" || nm == "neptune_ir.as_tensor" ||
                  nm == "neptune_ir.from_tensor" || nm == "arith.constant")) clean = false;
        }
        if (clean) returned_producer = p;
      }
    }

    std::ostringstream o;
    const std::string impl = f.name + "__impl";
    o << "// ---- @" << f.name << " (line " << f.line << ") ----\n";
    o << "static nl::Val " << impl << "(nl::Scope& sc";
    for (int i = 0; i < nargs; ++i) o << ", const nl::Val& " << cname(f.body.args[i].name);
    o << ", const nl::Val* dest, int* ret_arg, double* sret) {\n";
    o << "  (void)dest; (void)sret; if (ret_arg) *ret_arg = -1;\n";
    if (returned_producer >= 0 && nargs > 0) {
      // ... and never when it overlaps one of this function's own arguments (checked on the actual pointers): the
      // producer would overwrite data the function was given to read
      o << "  if (dest && (";
      for (int i = 0; i < nargs; ++i) o << (i ? " || " : "") << "nl::overlaps(*dest, " << cname(f.body.args[i].name) << ")";
      o << ")) dest = nullptr;\n";
    }
    int apply_counter = 0;
    for (size_t oi = 0; oi < f.body.ops.size(); ++oi) {
      const Op& op = *f.body.ops[oi];
      const std::string& n = op.name;
      if (n == "neptune_ir.wrap" || n == "neptune_ir.unwrap" || n == "neptune_ir.load" || n == "neptune_ir.as_tensor" ||
          n == "neptune_ir.from_tensor") {
        ValueInfo vi;
        vi.type = op.types[1];
        vi.root_arg = vals[op.operands[0]].root_arg;
        vi.uses = vals[op.results[0]].uses;
        vals[op.results[0]] = vi;
        Bounds b;
        if (op.types[1].is_tempish()) b = op.types[1].bounds;
        o << "  // " << n << " " << op.operands[0] << "\n";
        if (op.types[1].is_tempish()) {
          o << "  const nl::Val " << cname(op.results[0]) << " = sc.alias(" << cname(op.operands[0]) << ", " << new_box(b)
            << ", \"" << n << "\");\n";
        } else {  // unwrap -> memref: same buffer, zero-based box
          Bounds zb = op.types[0].bounds;
          for (int d = 0; d < zb.rank(); ++d) { zb.ub[d] -= zb.lb[d]; zb.lb[d] = 0; }
          o << "  const nl::Val " << cname(op.results[0]) << " = sc.alias(" << cname(op.operands[0]) << ", " << new_box(zb)
            << ", \"" << n << "\");\n";
        }
      } else if (n == "neptune_ir.apply") {
        const std::string tag = f.name + "_" + std::to_string(apply_counter++);
        Footprint fp;
        if (!emit_body(op, tag, fp)) return false;
        const int nin = (int)op.operands.size();
        const Type& res = op.types[nin];
        ValueInfo vi;
        vi.type = res;
        vi.uses = vals[op.results[0]].uses;
        vals[op.results[0]] = vi;
        // single-use result consumed by a reduce a few scalar/alias ops later: the apply is evaluated
        // inside the reduction kernel (run_apply_reduce_sum), the temp never exists
        {
          int consumer = -1;
          if (vi.uses == 1 && fp.lead == 0)
            for (size_t j = oi + 1; j < f.body.ops.size(); ++j) {
              const Op& c = *f.body.ops[j];
              if (c.name == "neptune_ir.reduce" && c.operands.at(0) == op.results[0]) { consumer = (int)j; break; }
              const std::string& nm = c.name;
              const bool harmless = nm == "neptune_ir.wrap" || nm == "neptune_ir.unwrap" || nm == "neptune_ir.load" ||
                                    ((nm.compare(0, 6, "arith.") == 0 || nm.compare(0, 5, "math.") == 0) && c.regions.empty());
              if (!harmless) break;
            }
          if (consumer >= 0) {
            FusedReduce fr;
            fr.tag = tag;
            fr.elem = ctype(res.elem);
            fr.rank = res.bounds.rank();
            fr.nin = nin;
            fr.result_box = new_box(res.bounds);
            fr.bounds_box = new_box(op.attrs.at("bounds").bounds);
            fr.halo0 = halo0_of(fp);
            o << "  // neptune_ir.apply -> " << op.results[0] << "   (evaluated inside the reduce below)\n";
            o << "  const nl::Val* in_" << tag << "[] = {";
            for (int k = 0; k < nin; ++k) o << (k ? ", " : "") << "&" << cname(op.operands[k]);
            o << "};\n";
            fused_reduce[op.results[0]] = fr;
            ApplyInfo ai;
            ai.function = f.name;
            ai.tag = tag;
            ai.rank = fp.rank;
            ai.num_inputs = fp.nin;
            ai.march = false;
            ai.box = fp.box;
            ai.halo_input = fp.halo_inputs > 0 ? std::max(fp.halo_input, 0) : -1;  // report: star/box also when the direct kernel runs it
            ai.fused_reduce = true;
            ai.exact = fp.exact;
            info.applies.push_back(ai);
            continue;
          }
        }
        std::string dest = "nullptr";
        if (dest_of.count((int)oi)) dest = dest_of[(int)oi];
        else if ((int)oi == returned_producer) dest = "dest";
        o << "  // neptune_ir.apply -> " << op.results[0] << "   (kernel + fused copy-through)\n";
        o << "  const nl::Val* in_" << tag << "[] = {";
        for (int k = 0; k < nin; ++k) o << (k ? ", " : "") << "&" << cname(op.operands[k]);
        o << "};\n";
        if (fp.nd) {
          // rank 4..6 with offsets along a leading dimension: the rank-generic kernel (lowered_runtime.hpp run_apply_nd)
          o << "  const nl::Val " << cname(op.results[0]) << " = nl::run_apply_nd<Body_" << tag << ", " << ctype(res.elem) << ", "
            << res.bounds.rank() << ", " << nin << ">(sc, Body_" << tag << "{}, " << new_box(res.bounds) << ", "
            << new_box(op.attrs.at("bounds").bounds) << ", in_" << tag << ", kNdReach_" << tag << ", " << dest << ", " << fp.nd_halo0 << ");\n";
          ApplyInfo ai;
          ai.function = f.name;
          ai.tag = tag;
          ai.rank = res.bounds.rank();
          ai.num_inputs = fp.nin;
          ai.march = false;
          ai.box = true;
          ai.halo_input = 0;
          ai.elem = res.elem;
          ai.halo0 = fp.nd_halo0;
          ai.geom_symbol = "";     // no geometry-level entry: neptune_hip_apply_geom_t is rank 1..3
          ai.exact = fp.exact;
          info.applies.push_back(ai);
          continue;
        }
        if (fp.lead > 0) {
          // rank 4..6: one rank-3 apply per index of the leading dimensions (lowered_runtime.hpp run_apply_batched)
          o << "  const nl::Val " << cname(op.results[0]) << " = nl::run_apply_batched<Body_" << tag << ", " << ctype(res.elem) << ", "
            << res.bounds.rank() << ", " << nin << ", FP_" << tag << ">(sc, Body_" << tag << "{}, " << new_box(res.bounds) << ", "
            << new_box(op.attrs.at("bounds").bounds) << ", in_" << tag << ", kTopRadius_" << tag << ", " << dest << ");\n";
          ApplyInfo ai;
          ai.function = f.name;
          ai.tag = tag;
          ai.rank = res.bounds.rank();
          ai.num_inputs = fp.nin;
          ai.march = fp.march_ok;
          ai.box = fp.box;
          ai.halo_input = fp.halo_inputs > 0 ? std::max(fp.halo_input, 0) : -1;
          ai.elem = res.elem;
          ai.halo0 = 0;
          ai.geom_symbol = "";     // no geometry-level entry: neptune_hip_apply_geom_t is rank 1..3
          ai.exact = fp.exact;
          info.applies.push_back(ai);
          continue;
        }
        o << "  const nl::Val " << cname(op.results[0]) << " = nl::run_apply<Body_" << tag << ", " << ctype(res.elem) << ", "
          << res.bounds.rank() << ", " << nin << ", FP_" << tag << ">(sc, Body_" << tag << "{}, " << new_box(res.bounds) << ", "
          << new_box(op.attrs.at("bounds").bounds) << ", in_" << tag << ", kTopRadius_" << tag << ", " << dest << ", "
          << halo0_of(fp) << ");\n";
        ApplyInfo ai;
        ai.function = f.name;
        ai.tag = tag;
        ai.rank = fp.rank;
        ai.num_inputs = fp.nin;
        ai.march = fp.march_ok;
        ai.box = fp.box;
        ai.halo_input = fp.halo_inputs > 0 ? std::max(fp.halo_input, 0) : -1;  // report: star/box also when the direct kernel runs it
        ai.elem = res.elem;
        ai.halo0 = halo0_of(fp);
        ai.geom_symbol = tag + "__geom";
        ai.exact = fp.exact;
        // Geometry-level entry of this apply's body: what neptune_hip_apply_builtin is for the library's own
        // bodies (caller-supplied boxes, bounds, region, stream and launch configuration; no allocation,
        // no synchronisation).  The slab decomposition drives user stencils through it.
        geom_entries << "extern \"C\" int " << ai.geom_symbol
                     << "(const neptune_hip_apply_geom_t* g, const void* const* in, void* out, void* stream,\n"
                     << "    const neptune_hip_launch_cfg_t* cfg) {\n"
                     << "  if (!g || !in || !out) return NEPTUNE_HIP_EINVAL;\n"
                     << "  const int rc = neptune_hip::geom_check_radius(g, kTopRadius_" << tag << ");\n"
                     << "  if (rc != NEPTUNE_HIP_OK) return rc;\n"
                     << "  return neptune_hip::launch_apply<Body_" << tag << ", " << ctype(res.elem) << ", " << res.bounds.rank() << ", " << nin
                     << ", FP_" << tag << ">(Body_" << tag << "{}, g, in, out, (hipStream_t)stream, cfg);\n}\n"
                     << "// two chained applies of this body in one pass over HBM (csrc/kernels/apply_march2.hpp): out = A(A(in)), or\n"
                     << "// NEPTUNE_HIP_EUNSUPPORTED when the footprint / geometry does not qualify (neptune_hip_step_loop_pairs)\n"
                     << "extern \"C\" int " << ai.geom_symbol << "2"
                     << "(const neptune_hip_apply_geom_t* g, const void* const* in, void* out, void* stream,\n"
                     << "    const neptune_hip_launch_cfg_t* cfg) {\n"
                     << "  if (!g || !in || !out) return NEPTUNE_HIP_EINVAL;\n"
                     << "  const int rc = neptune_hip::geom_check_radius(g, kTopRadius_" << tag << ");\n"
                     << "  if (rc != NEPTUNE_HIP_OK) return rc;\n"
                     << "  return neptune_hip::launch_apply_twice<Body_" << tag << ", " << ctype(res.elem) << ", " << res.bounds.rank() << ", " << nin
                     << ", FP_" << tag << ">(Body_" << tag << "{}, g, in, out, (hipStream_t)stream, cfg);\n}\n"
                     << "extern \"C\" int " << ai.geom_symbol << "3"
                     << "(const neptune_hip_apply_geom_t* g, const void* const* in, void* out, void* stream,\n"
                     << "    const neptune_hip_launch_cfg_t* cfg) {\n"
                     << "  if (!g || !in || !out) return NEPTUNE_HIP_EINVAL;\n"
                     << "  const int rc = neptune_hip::geom_check_radius(g, kTopRadius_" << tag << ");\n"
                     << "  if (rc != NEPTUNE_HIP_OK) return rc;\n"
                     << "  return neptune_hip::launch_apply_thrice<Body_" << tag << ", " << ctype(res.elem) << ", " << res.bounds.rank() << ", " << nin
                     << ", FP_" << tag << ">(Body_" << tag << "{}, g, in, out, (hipStream_t)stream, cfg);\n}\n"
                     << "// march tiles this module holds for that entry (plan-time tuning: neptune_hip_autotune_fn)\n"
                     << "extern \"C\" int " << ai.geom_symbol << "_variants(int rank) { return neptune_hip::march_variant_count(rank); }\n\n";
        info.applies.push_back(ai);
      } else if (n == "neptune_ir.time_advance") {
        // explicit Euler step: k = rhs(state); result = state + dt * k, over the whole box.  The
        // reference's own explicit lowering (HighLevelConvertion.cpp:77-120) builds exactly this
        // apply_{linear,nonlinear} + axpy apply pair (its version is 1-D-only and ill-formed).
        const Type& st = op.types[0];
        ValueInfo vi;
        vi.type = st;
        vi.uses = vals[op.results[0]].uses;
        vals[op.results[0]] = vi;
        std::string dest = "nullptr";
        if (dest_of.count((int)oi)) dest = dest_of[(int)oi];
        else if ((int)oi == returned_producer) dest = "dest";
        const std::string tag = f.name + "_ta" + std::to_string(apply_counter++);
        const std::string bx = new_box(st.bounds);
        const std::string T = ctype(st.elem);
        // Fusable: the rhs opdef is exactly "apply(state) ; return" with result box == state box.  Then
        // one kernel computes state + dt * rhs(state) (ops::EulerFused); anything else takes the
        // two-kernel form below.  Both produce the same bits.
        const Function* c = m.find(op.callee);
        const Op* rhs_apply = nullptr;
        if (st.rank() <= 3 && c && c->arg_types.size() == 1 && c->body.ops.size() == 2 && c->body.ops[0]->name == "neptune_ir.apply" &&
            c->body.ops[1]->name == "neptune_ir.return" && c->body.ops[1]->operands.size() == 1 &&
            c->body.ops[1]->operands[0] == c->body.ops[0]->results.at(0) && c->body.ops[0]->operands.size() == 1 &&
            c->body.ops[0]->operands[0] == c->body.args[0].name) {
          const Op& a = *c->body.ops[0];
          const Type& rt = a.types[1];
          bool same = a.types[0].is_tempish() && rt.elem == st.elem && rt.bounds.rank() == st.rank() && a.types[0].elem == st.elem;
          for (int d = 0; same && d < st.rank(); ++d)
            same = rt.bounds.lb[d] == st.bounds.lb[d] && rt.bounds.ub[d] == st.bounds.ub[d] &&
                   a.types[0].bounds.lb[d] == st.bounds.lb[d] && a.types[0].bounds.ub[d] == st.bounds.ub[d];
          if (same) rhs_apply = &a;
        }
        if (rhs_apply) {
          Footprint cfp;
          std::map<std::string, int> ti, ia;
          if (!analyze_apply(*rhs_apply, cfp, ti, ia)) return false;
          const std::string ctag = op.callee + "_0";  // the tag emit_function gives the opdef's only apply
          o << "  // neptune_ir.time_advance {method = 0 (explicit), rhs = @" << op.callee
            << "}: state + dt * rhs(state), rhs apply and axpy fused into one kernel\n";
          o << "  static const neptune_hip::Reach kTopRadius_" << tag << " = " << fused_reach(cfp, st.rank()) << ";\n";
          o << "  const nl::Val* in_" << tag << "[] = {&" << cname(op.operands[0]) << "};\n";
          const std::string body = "neptune_hip::ops::EulerFused<Body_" + ctag + ", " + T + ", " + std::to_string(st.rank()) + ">";
          o << "  const nl::Val " << cname(op.results[0]) << " = nl::run_apply<" << body << ", " << T << ", " << st.rank() << ", 1, FP_"
            << ctag << ">(sc, " << body << "{(" << T << ")" << cname(op.operands[1]) << "}, " << bx << ", "
            << new_box(rhs_apply->attrs.at("bounds").bounds) << ", in_" << tag << ", kTopRadius_" << tag << ", " << dest << ", "
            << halo0_of(cfp) << ");\n";
          ApplyInfo ai;
          ai.function = f.name;
          ai.tag = tag;
          ai.rank = st.rank();
          ai.num_inputs = 1;
          ai.march = cfp.march_ok;
          ai.box = cfp.box;
          ai.halo_input = cfp.halo_inputs > 0 ? std::max(cfp.halo_input, 0) : -1;
          ai.elem = st.elem;
          ai.halo0 = halo0_of(cfp);
          ai.exact = cfp.exact;
          // With a constant time step the fused step is a self-contained apply: give it geometry-level entries too
          // (single step, two steps per pass, tile count), so step loops and slab decompositions can drive `u + dt*rhs(u)`
          // exactly like a plain operator.
          {
            auto dit = def_at.find(op.operands[1]);
            const Op* dtdef = dit == def_at.end() ? nullptr : f.body.ops[dit->second].get();
            bool lit_ok = false;
            std::string lit;
            if (dtdef && dtdef->name == "arith.constant") lit = float_literal(dtdef->literal, st.elem, lit_ok);
            if (lit_ok) {
              ai.geom_symbol = tag + "__geom";
              std::ostringstream& ge = geom_entries;
              ge << "static const neptune_hip::Reach kTopRadiusG_" << tag << " = " << fused_reach(cfp, st.rank()) << ";\n";
              const char* names[3] = {"", "2", "3"};
              const char* fns[3] = {"launch_apply", "launch_apply_twice", "launch_apply_thrice"};
              for (int v = 0; v < 3; ++v)
                ge << "extern \"C\" int " << ai.geom_symbol << names[v]
                   << "(const neptune_hip_apply_geom_t* g, const void* const* in, void* out, void* stream,\n"
                   << "    const neptune_hip_launch_cfg_t* cfg) {\n"
                   << "  if (!g || !in || !out) return NEPTUNE_HIP_EINVAL;\n"
                   << "  const int rc = neptune_hip::geom_check_radius(g, kTopRadiusG_" << tag << ");\n"
                   << "  if (rc != NEPTUNE_HIP_OK) return rc;\n"
                   << "  return neptune_hip::" << fns[v] << "<" << body << ", " << T << ", " << st.rank() << ", 1, FP_" << ctag << ">(" << body
                   << "{(" << T << ")" << lit << "}, g, in, out, (hipStream_t)stream, cfg);\n}\n";
              ge << "extern \"C\" int " << ai.geom_symbol << "_variants(int rank) { return neptune_hip::march_variant_count(rank); }\n\n";
            }
          }
          info.applies.push_back(ai);
        } else {
          o << "  // neptune_ir.time_advance {method = 0 (explicit), rhs = @" << op.callee << "}: state + dt * rhs(state)\n";
          o << "  const nl::Val k_" << tag << " = " << op.callee << "__impl(sc, " << cname(op.operands[0]) << ", nullptr, nullptr, nullptr);\n";
          if (st.rank() > 3) {
            // rank 4..6: the rhs through its own lowering (leading dimensions peeled off, or the rank-generic kernel), the
            // axpy as one flat pointwise pass
            o << "  const nl::Val " << cname(op.results[0]) << " = nl::run_euler_axpy_flat<" << T << ">(sc, (" << T << ")" << cname(op.operands[1])
              << ", " << cname(op.operands[0]) << ", k_" << tag << ", " << dest << ");\n";
            ApplyInfo ai;
            ai.function = f.name;
            ai.tag = tag;
            ai.rank = st.rank();
            ai.num_inputs = 2;
            ai.march = true;
            ai.halo_input = -1;
            info.applies.push_back(ai);
            continue;
          }
          o << "  const nl::Val* in_" << tag << "[] = {&" << cname(op.operands[0]) << ", &k_" << tag << "};\n";
          o << "  const nl::Val " << cname(op.results[0]) << " = nl::run_apply<neptune_hip::ops::EulerAxpy<" << T << ", " << st.rank()
            << ">, " << T << ", " << st.rank() << ", 2, nl::PointwiseFP>(sc, neptune_hip::ops::EulerAxpy<" << T << ", " << st.rank() << ">{(" << T << ")"
            << cname(op.operands[1]) << "}, " << bx << ", " << bx << ", in_" << tag << ", nl::kPointwiseRadius2, " << dest << ");\n";
          ApplyInfo ai;
          ai.function = f.name;
          ai.tag = tag;
          ai.rank = st.rank();
          ai.num_inputs = 2;
          ai.march = true;
          ai.halo_input = -1;
          info.applies.push_back(ai);
        }
      } else if (!op.callee.empty()) {
        const Function* c = m.find(op.callee);
        ValueInfo vi;
        vi.type = c->result_types.at(0);
        vi.uses = vals[op.results[0]].uses;
        vals[op.results[0]] = vi;
        std::string dest = "nullptr";
        if (dest_of.count((int)oi)) dest = dest_of[(int)oi];
        else if ((int)oi == returned_producer) dest = "dest";
        o << "  // " << n << " @" << op.callee << "\n";
        o << "  const nl::Val " << cname(op.results[0]) << " = " << op.callee << "__impl(sc";
        for (auto& a : op.operands) o << ", " << cname(a);
        o << ", " << dest << ", nullptr, nullptr);\n";
      } else if (n == "neptune_ir.store") {
        const Type& vt = op.types[0];
        o << "  // neptune_ir.store " << op.operands[0] << " to " << op.operands[1] << "\n";
        if (op.attrs.count("bounds")) {
          std::string bx = new_box(op.attrs.at("bounds").bounds);
          o << "  nl::run_store(sc, " << cname(op.operands[0]) << ", " << cname(op.operands[1]) << ", &" << bx << ", "
            << dtype_macro(vt.elem) << ");\n";
        } else {
          o << "  nl::run_store(sc, " << cname(op.operands[0]) << ", " << cname(op.operands[1]) << ", nullptr, "
            << dtype_macro(vt.elem) << ");\n";
        }
      } else if (n == "neptune_ir.reduce") {
        scalar_kind[op.results[0]] = 1;   // under a slab view: this rank's partial sum
        const Type& in = op.types[0];
        std::string bx = "nullptr";
        if (op.attrs.count("bounds")) bx = "&" + new_box(op.attrs.at("bounds").bounds);
        auto fit = fused_reduce.find(op.operands[0]);
        if (fit != fused_reduce.end()) {
          const FusedReduce& fr = fit->second;
          o << "  // neptune_ir.reduce " << op.operands[0] << " {kind = \"sum\"}   (apply + fixed-tree device sum in one kernel, blocking)\n";
          o << "  const " << fr.elem << " " << cname(op.results[0]) << " = (" << fr.elem << ")nl::run_apply_reduce_sum<Body_" << fr.tag
            << ", " << fr.elem << ", " << fr.rank << ", " << fr.nin << ", FP_" << fr.tag << ">(sc, Body_" << fr.tag << "{}, "
            << fr.result_box << ", " << fr.bounds_box << ", in_" << fr.tag << ", kTopRadius_" << fr.tag << ", " << fr.halo0 << ", "
            << bx << ");\n";
          continue;
        }
        o << "  // neptune_ir.reduce " << op.operands[0] << " {kind = \"sum\"}   (fixed-tree device sum, blocking)\n";
        o << "  const " << ctype(in.elem) << " " << cname(op.results[0]) << " = (" << ctype(in.elem) << ")nl::run_reduce_sum(sc, "
          << cname(op.operands[0]) << ", " << bx << ", " << dtype_macro(in.elem) << ");\n";
      } else if ((n.compare(0, 6, "arith.") == 0 || n.compare(0, 5, "math.") == 0) && op.regions.empty()) {
        // scalar arithmetic at function level (constants, reduce results): plain host statements
        static const std::map<std::string, int> none;
        if (!emit_op(op, o, "  ", none, none, nullptr)) return false;
        // what a scalar means when the function runs on one slab of a decomposed field: 0 = the same on every rank
        // (constants and arithmetic on them), 1 = a bare reduce result (ranks add up), 2 = computed FROM a partial sum
        // (sqrt of it, a product of two, ...): not recoverable from the per-rank values
        int kind = 0;
        for (auto& a : op.operands) {
          auto it = scalar_kind.find(a);
          if (it != scalar_kind.end() && it->second != 0) kind = 2;
        }
        for (auto& r : op.results) scalar_kind[r] = kind;
      } else if (n == "neptune_ir.return" || n == "func.return" || n == "return") {
        if (op.operands.empty()) {
          o << "  return nl::Val{};\n";
        } else if (f.result_types[0].is_scalar()) {
          auto it = scalar_kind.find(op.operands[0]);
          const int kind = it == scalar_kind.end() ? 0 : it->second;
          returned_scalar_kind = (returned_scalar_kind < 0 || returned_scalar_kind == kind) ? kind : 2;
          if (kind == 2)
            o << "  if (sc.has_ghosts()) nl::die(\"" << f.name << "\", \"slab mode: the returned scalar is computed from a reduce result, "
              << "which is only this rank's partial sum -- return the bare reduce and finish the arithmetic after the ranks' sums are added\");\n";
          o << "  if (sret) *sret = (double)" << cname(op.operands[0]) << ";\n  return nl::Val{};\n";
        } else {
          const int root = vals[op.operands[0]].root_arg;
          if (root >= 0) o << "  if (ret_arg) *ret_arg = " << root << ";\n";
          o << "  return " << cname(op.operands[0]) << ";\n";
        }
      }
    }
    o << "}\n";

    // exported symbol with the expanded-memref ABI
    const bool has_res = !f.result_types.empty();
    const bool scalar_res = has_res && f.result_types[0].is_scalar();
    const int rrank = (has_res && !scalar_res) ? f.result_types[0].rank() : 0;
    o << "extern \"C\" "
      << (scalar_res ? ctype(f.result_types[0].elem) : has_res ? "NeptuneMemRef" + std::to_string(rrank) + "D" : std::string("void"))
      << " " << f.name << "(";
    for (int i = 0; i < nargs; ++i) {
      const int r = f.arg_types[i].rank();
      const std::string a = "a" + std::to_string(i);
      o << (i ? ", " : "") << "void* " << a << "_allocated, void* " << a << "_aligned, int64_t " << a << "_offset";
      for (int d = 0; d < r; ++d) o << ", int64_t " << a << "_size" << d;
      for (int d = 0; d < r; ++d) o << ", int64_t " << a << "_stride" << d;
    }
    o << ") {\n";
    o << "  nl::Scope sc(\"" << f.name << "\");\n";
    for (int i = 0; i < nargs; ++i) {
      const Type& t = f.arg_types[i];
      const int r = t.rank();
      const std::string a = "a" + std::to_string(i);
      o << "  const int64_t " << a << "_sizes[] = {";
      for (int d = 0; d < r; ++d) o << (d ? ", " : "") << a << "_size" << d;
      o << "}, " << a << "_strides[] = {";
      for (int d = 0; d < r; ++d) o << (d ? ", " : "") << a << "_stride" << d;
      o << "};\n";
      o << "  nl::Val m" << i << " = sc.bind_memref(" << r << ", " << esize(t.elem) << ", " << a << "_allocated, " << a
        << "_aligned, " << a << "_offset, " << a << "_sizes, " << a << "_strides);\n";
      if (t.is_tempish()) {
        // opdef arguments are temps of static shape: the descriptor must match (memref.cast ?->static)
        o << "  m" << i << " = sc.alias(m" << i << ", " << new_box(t.bounds) << ", \"argument " << i << "\");\n";
      } else {
        for (int d = 0; d < r; ++d)
          if (t.shape[d] >= 0)
            o << "  if (" << a << "_size" << d << " != " << t.shape[d] << ") nl::die(\"" << f.name << "\", \"memref argument " << i
              << " has the wrong static extent\");\n";
      }
    }
    o << "  int ret_arg = -1;\n  double sret = 0.0;\n";
    o << "  " << ((has_res && !scalar_res) ? "const nl::Val r = " : "") << impl << "(sc";
    for (int i = 0; i < nargs; ++i) o << ", m" << i;
    o << ", nullptr, &ret_arg, &sret);\n";
    o << "  sc.finish();\n";
    if (scalar_res) {
      o << "  return (" << ctype(f.result_types[0].elem) << ")sret;\n";
    } else if (has_res) {
      const std::string mr = "NeptuneMemRef" + std::to_string(rrank) + "D";
      o << "  " << mr << " out;\n";
      // result aliases an argument (e.g. @entry returns unwrap of its destination field): hand the
      // caller's own descriptor back, as the reference does (smoke_apply.mlir:23-25)
      o << "  switch (ret_arg) {\n";
      for (int i = 0; i < nargs; ++i) {
        if (f.arg_types[i].rank() != rrank) continue;
        const std::string a = "a" + std::to_string(i);
        o << "    case " << i << ": out.allocated = " << a << "_allocated; out.aligned = " << a << "_aligned; out.offset = " << a
          << "_offset;";
        for (int d = 0; d < rrank; ++d) o << " out.sizes[" << d << "] = " << a << "_size" << d << "; out.strides[" << d << "] = " << a << "_stride" << d << ";";
        o << " return out;\n";
      }
      o << "    default: break;\n  }\n";
      o << "  void* p = sc.export_result(r);\n";
      o << "  return nl::make_memref<" << rrank << ">(p, r.box);\n";
    }
    o << "}\n\n";
    funcs << o.str();
    info.lowered.push_back(f.name);
    Signature sig;
    sig.name = f.name;
    auto conv = [](const Type& t) {
      SigType st;
      st.kind = t.kind == TypeKind::MemRef ? "memref" : (t.kind == TypeKind::Temp ? "temp" : (t.is_scalar() ? "scalar" : "field"));
      st.elem = t.elem;
      st.rank = t.rank();
      if (t.kind == TypeKind::MemRef) st.shape = t.shape;
      else if (t.is_tempish()) {
        for (int d = 0; d < t.rank(); ++d) { st.shape.push_back(t.bounds.ub[d] - t.bounds.lb[d]); st.lb.push_back(t.bounds.lb[d]); }
      }
      return st;
    };
    for (auto& t : f.arg_types) sig.args.push_back(conv(t));
    sig.has_result = has_res;
    if (has_res) sig.result = conv(f.result_types[0]);
    if (scalar_res) sig.result.scalar = returned_scalar_kind <= 0 ? "uniform" : (returned_scalar_kind == 1 ? "partial_sum" : "derived");
    info.signatures.push_back(sig);
    return true;
  }

  // ---- outlining: the stencil part of a function that also holds solver ops ----------------------------------
  // Every current-style input of the reference has this shape (test/smoke_tests/smoke_time_advance.mlir:53-84: @entry
  // computes %ustar with a neptune_ir.apply, then hands it to an implicit time_advance).  The solver op stays on the
  // host (RuntimeLowering, out of scope); what CAN run on the GPU is every value such an op consumes that is computed
  // by applies / operator calls from the function's own arguments.  Each becomes an exported symbol
  // <function>__stencil_<k>(<the function's memref arguments>) -> memref, with the expanded-memref ABI like any
  // lowered function, which the reference's CPU-lowered function can call in place of its scf loop nest.
  static std::unique_ptr<Op> clone_op(const Op& src) {
    auto op = std::make_unique<Op>();
    op->name = src.name; op->results = src.results; op->operands = src.operands; op->attrs = src.attrs; op->types = src.types;
    op->offsets = src.offsets; op->callee = src.callee; op->literal = src.literal; op->predicate = src.predicate;
    op->opaque = src.opaque; op->line = src.line;
    for (auto& r : src.regions) {
      auto b = std::make_unique<Block>();
      b->args = r->args;
      for (auto& o : r->ops) b->ops.push_back(clone_op(*o));
      op->regions.push_back(std::move(b));
    }
    return op;
  }
  std::vector<std::unique_ptr<Function>> outlined_funcs;   // keeps the synthetic functions alive

  void outline_stencil_parts(const Function& f, std::vector<const Function*>& todo) {
    const auto& ops = f.body.ops;
    std::map<std::string, int> def_at;
    for (size_t oi = 0; oi < ops.size(); ++oi)
      for (auto& r : ops[oi]->results) def_at[r] = (int)oi;
    // tainted: results of solver ops and everything computed from them -- and, from a store of such a value on, the
    // field it went into (and every alias of that field)
    std::map<std::string, std::string> root;   // value -> the function argument / value it aliases
    auto root_of = [&](const std::string& v) { auto it = root.find(v); return it == root.end() ? v : it->second; };
    std::map<std::string, bool> tainted;
    std::map<std::string, bool> dirty_root;
    std::vector<bool> op_tainted(ops.size(), false);
    for (size_t oi = 0; oi < ops.size(); ++oi) {
      const Op& op = *ops[oi];
      const std::string& n = op.name;
      if (n == "neptune_ir.wrap" || n == "neptune_ir.unwrap" || n == "neptune_ir.load" || n == "neptune_ir.as_tensor" || n == "neptune_ir.from_tensor")
        if (!op.results.empty() && !op.operands.empty()) root[op.results[0]] = root_of(op.operands[0]);
      bool t = op.opaque;
      for (auto& v : op.operands) t = t || tainted[v] || dirty_root[root_of(v)];
      op_tainted[oi] = t;
      for (auto& r : op.results) tainted[r] = t;
      if (t && n == "neptune_ir.store" && op.operands.size() >= 2) dirty_root[root_of(op.operands[1])] = true;
    }
    // live-outs: untainted temps produced by an apply / operator call / explicit time_advance and consumed by a tainted op
    std::vector<std::string> live;
    for (size_t oi = 0; oi < ops.size(); ++oi) {
      if (!op_tainted[oi]) continue;
      for (auto& v : ops[oi]->operands) {
        auto d = def_at.find(v);
        if (d == def_at.end() || op_tainted[d->second]) continue;
        const Op& p = *ops[d->second];
        const bool producer = p.name == "neptune_ir.apply" || !p.callee.empty();
        if (producer && std::find(live.begin(), live.end(), v) == live.end()) live.push_back(v);
      }
    }
    int k = 0;
    for (const std::string& v : live) {
      // backward slice of v over untainted ops
      std::vector<bool> need(ops.size(), false);
      std::vector<std::string> work{v};
      bool ok = true;
      while (!work.empty()) {
        const std::string x = work.back();
        work.pop_back();
        auto d = def_at.find(x);
        if (d == def_at.end()) continue;            // a function argument
        if (op_tainted[d->second]) { ok = false; break; }
        if (need[d->second]) continue;
        need[d->second] = true;
        for (auto& o : ops[d->second]->operands) work.push_back(o);
      }
      const int last = def_at[v];
      // a store before the producer could feed the slice through memory (a load of the stored field): such functions
      // are left alone rather than outlined with a side effect the caller would see twice
      for (int oi = 0; ok && oi < last; ++oi)
        if (ops[oi]->name == "neptune_ir.store") ok = false;
      if (!ok) continue;
      const Op& prod = *ops[last];
      const Type vt = prod.name == "neptune_ir.apply" ? prod.types[prod.operands.size()] : (prod.name == "neptune_ir.time_advance" ? prod.types[0] : prod.types.back());
      if (!vt.is_tempish()) continue;
      auto g = std::make_unique<Function>();
      g->name = f.name + "__stencil_" + std::to_string(k);
      g->kind = FuncKind::Func;
      g->arg_types = f.arg_types;
      g->result_types = {vt};
      g->line = prod.line;
      g->body.args = f.body.args;
      for (size_t oi = 0; oi < ops.size(); ++oi)
        if (need[oi]) g->body.ops.push_back(clone_op(*ops[oi]));
      auto ret = std::make_unique<Op>();
      ret->name = "func.return";
      ret->operands = {v};
      ret->types = {vt};
      ret->line = prod.line;
      g->body.ops.push_back(std::move(ret));
      std::string why;
      if (!lowerable(*g, why)) continue;
      info.outlined.push_back({g->name, f.name, v, prod.line});
      todo.push_back(g.get());
      outlined_funcs.push_back(std::move(g));
      ++k;
    }
  }

  // FNV-1a of the emitted constants and body functors: two modules with the same bodies share their measured launches
  static std::string module_id(const std::string& text) {
    unsigned long long h = 1469598103934665603ull;
    for (unsigned char c : text) { h ^= c; h *= 1099511628211ull; }
    char buf[32];
    snprintf(buf, sizeof buf, "%016llx", h);
    return buf;
  }

  bool run(std::string& out) {
    // callees before callers: opdefs are emitted in module order, forward declarations cover the rest
    std::ostringstream fwd;
    std::vector<const Function*> todo;
    for (auto& f : m.funcs) {
      std::string why;
      if (!lowerable(*f, why)) {
        info.skipped.push_back({f->name, why});
        outline_stencil_parts(*f, todo);    // ... but the stencil values its solver ops consume are (see above)
        continue;
      }
      todo.push_back(f.get());
    }
    for (auto* f : todo) {
      fwd << "static nl::Val " << f->name << "__impl(nl::Scope& sc";
      for (size_t i = 0; i < f->arg_types.size(); ++i) fwd << ", const nl::Val&";
      fwd << ", const nl::Val* dest, int* ret_arg, double* sret);\n";
    }
    for (auto* f : todo)
      if (!emit_function(*f)) return false;
    std::ostringstream o;
    o << "// Generated by the NeptuneIR HIP lowering (neptune-opt --neptuneir-to-hip).  Do not edit.\n"
      << "// Compile: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared <this file> \\\n"
      << "//          -I<repo> -L<repo>/neptune-pde-solver_amd/lib -lneptune_hip\n"
      << "// identifies this module's body functors in the launch-wisdom keys (include/neptune_hip.h)\n"
      << "#define NEPTUNE_HIP_MODULE_ID \"m" << module_id(consts.str() + bodies.str()) << "\"\n"
      << "#include \"neptune-pde-solver_amd/csrc/runtime/lowered_runtime.hpp\"\n"
      << "#include \"neptune-pde-solver_amd/csrc/kernels/body_ops.hpp\"\n\n"
      << "namespace nl = neptune_hip::lowered;\n\nnamespace {\n\n"
      << consts.str() << "\n"
      << bodies.str() << "}  // namespace\n\n"
      << fwd.str() << "\n"
      << funcs.str()
      << "// ---- geometry-level entries (one per apply; see include/neptune_hip.h neptune_hip_apply_builtin) ----\n"
      << geom_entries.str();
    for (auto& s : info.skipped) o << "// not lowered: @" << s.first << ": " << s.second << "\n";
    out = o.str();
    return true;
  }
};

}  // namespace

bool lower_to_hip(const Module& m, std::string& out_source, LowerInfo& info, Diag& diag) {
  Emitter e(m, diag, info);
  return e.run(out_source) && diag.ok;
}

}  // namespace neptune_lowering

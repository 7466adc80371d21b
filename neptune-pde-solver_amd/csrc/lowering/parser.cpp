// parser.cpp -- recursive-descent parser for the textual NeptuneIR subset described in ir.h.
// Accepts exactly the assembly formats of the reference's ODS (NeptuneIROps.td assemblyFormat
// strings, cited per op below) plus attribute/type aliases (`#b = ...`, `!temp = ...`) as used by
// the reference's inputs (test/smoke_tests/smoke_time_advance.mlir:3-7).
#include <cctype>
#include <cstdlib>
#include <cstring>

#include "ir.h"

namespace neptune_lowering {

std::string Type::str() const {
  auto list = [](const std::vector<int64_t>& v) {
    std::string s;
    for (size_t i = 0; i < v.size(); ++i) s += (i ? ", " : "") + std::to_string(v[i]);
    return s;
  };
  switch (kind) {
    case TypeKind::Scalar: return elem;
    case TypeKind::Temp:
    case TypeKind::Field:
      return std::string("!neptune_ir.") + (kind == TypeKind::Temp ? "temp" : "field") + "<element = " + elem +
             ", bounds = #neptune_ir.bounds<lb = [" + list(bounds.lb) + "], ub = [" + list(bounds.ub) +
             "]>, location = #neptune_ir.location<\"" + location + "\">>";
    case TypeKind::MemRef: {
      std::string s = tensor ? "tensor<" : "memref<";
      for (auto d : shape) s += (d < 0 ? std::string("?") : std::to_string(d)) + "x";
      return s + elem + ">";
    }
    default: return "<none>";
  }
}

namespace {

enum class Tk { Id, Num, Str, Punct, Arrow, MemRef, Eof };
struct Tok {
  Tk kind;
  std::string text;
  int line;
};

bool is_id_start(char c) { return std::isalpha((unsigned char)c) || c == '_'; }
bool is_id_char(char c) { return std::isalnum((unsigned char)c) || c == '_' || c == '.' || c == '$'; }

bool tokenize(const std::string& s, std::vector<Tok>& out, Diag& diag) {
  size_t i = 0, n = s.size();
  int line = 1;
  while (i < n) {
    char c = s[i];
    if (c == '\n') { ++line; ++i; continue; }
    if (std::isspace((unsigned char)c)) { ++i; continue; }
    if (c == '/' && i + 1 < n && s[i + 1] == '/') {
      while (i < n && s[i] != '\n') ++i;
      continue;
    }
    if (c == '-' && i + 1 < n && s[i + 1] == '>') { out.push_back({Tk::Arrow, "->", line}); i += 2; continue; }
    if (c == '"') {
      size_t j = i + 1;
      std::string v;
      while (j < n && s[j] != '"') {
        if (s[j] == '\\' && j + 1 < n) ++j;
        v += s[j++];
      }
      if (j >= n) { diag.fail(line, "unterminated string"); return false; }
      out.push_back({Tk::Str, v, line});
      i = j + 1;
      continue;
    }
    bool sign = (c == '-' || c == '+') && i + 1 < n && (std::isdigit((unsigned char)s[i + 1]) || s[i + 1] == '.');
    if (std::isdigit((unsigned char)c) || sign || (c == '.' && i + 1 < n && std::isdigit((unsigned char)s[i + 1]))) {
      size_t j = i + (sign ? 1 : 0);
      if (j + 1 < n && s[j] == '0' && (s[j + 1] == 'x' || s[j + 1] == 'X')) {
        j += 2;
        while (j < n && std::isxdigit((unsigned char)s[j])) ++j;
      } else {
        while (j < n && std::isdigit((unsigned char)s[j])) ++j;
        if (j < n && s[j] == '.') { ++j; while (j < n && std::isdigit((unsigned char)s[j])) ++j; }
        if (j < n && (s[j] == 'e' || s[j] == 'E')) {
          size_t k = j + 1;
          if (k < n && (s[k] == '+' || s[k] == '-')) ++k;
          if (k < n && std::isdigit((unsigned char)s[k])) { j = k; while (j < n && std::isdigit((unsigned char)s[j])) ++j; }
        }
      }
      out.push_back({Tk::Num, s.substr(i, j - i), line});
      i = j;
      continue;
    }
    if (c == '%' || c == '@' || c == '^' || c == '#' || c == '!' || is_id_start(c)) {
      size_t j = i;
      bool sigil = !is_id_start(c);
      if (sigil) ++j;
      if (j < n && (is_id_start(s[j]) || (sigil && std::isdigit((unsigned char)s[j])))) {
        while (j < n && is_id_char(s[j])) ++j;
        std::string id = s.substr(i, j - i);
        // `tensor<4x8xf64>` (the operand / result of neptune_ir.as_tensor / from_tensor) is read like a static memref:
        // after bufferization a ranked tensor IS a dense buffer, and the reference keeps the two ops as casts
        // (lib/Passes/DataflowLowering.cpp:705-733)
        if (id == "memref" || id == "tensor") {
          size_t k = j;
          while (k < n && std::isspace((unsigned char)s[k])) ++k;
          if (k < n && s[k] == '<') {
            int depth = 0;
            size_t m = k;
            for (; m < n; ++m) {
              if (s[m] == '<') ++depth;
              else if (s[m] == '>' && --depth == 0) break;
            }
            if (m >= n) { diag.fail(line, "unterminated memref type"); return false; }
            out.push_back({Tk::MemRef, (id == "tensor" ? "T:" : "") + s.substr(k + 1, m - k - 1), line});
            i = m + 1;
            continue;
          }
        }
        out.push_back({Tk::Id, id, line});
        i = j;
        continue;
      }
    }
    if (std::strchr("{}()[]<>,:=?*", c)) { out.push_back({Tk::Punct, std::string(1, c), line}); ++i; continue; }
    diag.fail(line, std::string("unexpected character '") + c + "'");
    return false;
  }
  out.push_back({Tk::Eof, "", line});
  return true;
}

struct Parser {
  std::vector<Tok> t;
  size_t p = 0;
  Diag& diag;
  std::map<std::string, AttrValue> attr_alias;
  std::map<std::string, Type> type_alias;
  explicit Parser(Diag& d) : diag(d) {}

  const Tok& peek(size_t k = 0) const { return t[std::min(p + k, t.size() - 1)]; }
  const Tok& next() { const Tok& r = t[std::min(p, t.size() - 1)]; if (p < t.size() - 1) ++p; return r; }
  bool is(const char* s, size_t k = 0) const { return peek(k).kind != Tk::Str && peek(k).text == s; }
  bool accept(const char* s) { if (is(s)) { next(); return true; } return false; }
  bool expect(const char* s) {
    if (accept(s)) return true;
    diag.fail(peek().line, std::string("expected '") + s + "', got '" + peek().text + "'");
    return false;
  }
  bool ok() const { return diag.ok; }
  // loop condition of every delimited list: false once `close` has been consumed, on an earlier error, or --
  // with a diagnostic -- at the end of the input (a truncated file must not spin here)
  bool until(const char* close) {
    if (!ok()) return false;
    if (accept(close)) return false;
    if (peek().kind == Tk::Eof) {
      diag.fail(peek().line, std::string("unexpected end of input, expected '") + close + "'");
      return false;
    }
    return true;
  }
  bool is_value(size_t k = 0) const { return peek(k).kind == Tk::Id && peek(k).text[0] == '%'; }

  bool parse_int_list(std::vector<int64_t>& v) {
    if (!expect("[")) return false;
    while (until("]")) {
      if (peek().kind != Tk::Num) { diag.fail(peek().line, "expected integer, got '" + peek().text + "'"); return false; }
      v.push_back(std::strtoll(next().text.c_str(), nullptr, 0));
      accept(",");
    }
    return ok();
  }
  bool parse_bounds_body(Bounds& b) {  // after '#neptune_ir.bounds': `<` `lb` `=` $lb `,` `ub` `=` $ub `>`
    if (!expect("<")) return false;
    bool has_lb = false, has_ub = false;
    while (until(">")) {
      std::string key = next().text;
      if (!expect("=")) return false;
      if (key == "lb") { has_lb = true; if (!parse_int_list(b.lb)) return false; }
      else if (key == "ub") { has_ub = true; if (!parse_int_list(b.ub)) return false; }
      else { diag.fail(peek().line, "unknown bounds key '" + key + "'"); return false; }
      accept(",");
    }
    if (!has_lb || !has_ub || b.lb.size() != b.ub.size()) { diag.fail(peek().line, "bounds lb/ub rank mismatch"); return false; }
    // coordinates far beyond any buffer would overflow the extent arithmetic downstream (ub - lb, products of extents)
    const int64_t lim = (int64_t)1 << 40;
    long double cells = 1.0L;
    for (size_t d = 0; d < b.lb.size(); ++d) {
      if (b.lb[d] < -lim || b.lb[d] > lim || b.ub[d] < -lim || b.ub[d] > lim) {
        diag.fail(peek().line, "bounds coordinate outside the supported range [-2^40, 2^40]");
        return false;
      }
      if (b.ub[d] > b.lb[d]) cells *= (long double)(b.ub[d] - b.lb[d]);
    }
    if (cells > 4.0e18L) { diag.fail(peek().line, "bounds describe more than 2^62 cells"); return false; }
    return ok();
  }
  bool parse_attr_value(AttrValue& a) {
    const Tok& k = peek();
    if (k.text == "#neptune_ir.bounds") { next(); a.kind = AttrValue::BoundsK; return parse_bounds_body(a.bounds); }
    if (k.text == "#neptune_ir.location") {
      next();
      if (!expect("<")) return false;
      a.kind = AttrValue::String;
      a.s = next().text;
      return expect(">");
    }
    if (k.kind == Tk::Id && k.text[0] == '#') {
      auto it = attr_alias.find(k.text);
      if (it == attr_alias.end()) { diag.fail(k.line, "unknown attribute alias " + k.text); return false; }
      a = it->second;
      next();
      return true;
    }
    if (k.kind == Tk::Id && k.text[0] == '@') { a.kind = AttrValue::Symbol; a.s = next().text.substr(1); return true; }
    if (k.kind == Tk::Str) { a.kind = AttrValue::String; a.s = next().text; return true; }
    if (k.kind == Tk::Num) {
      std::string lit = next().text;
      bool isf = lit.find_first_of(".eE") != std::string::npos && lit.compare(0, 2, "0x") != 0;
      a.s = lit;
      if (isf) { a.kind = AttrValue::Float; a.f = std::strtod(lit.c_str(), nullptr); }
      else { a.kind = AttrValue::Int; a.i = std::strtoll(lit.c_str(), nullptr, 0); }
      if (accept(":")) { Type ty; if (!parse_type(ty)) return false; }
      return true;
    }
    if (k.text == "true" || k.text == "false") { a.kind = AttrValue::Bool; a.b = next().text == "true"; return true; }
    if (k.text == "[") {  // generic array attribute: parsed and dropped
      next();
      while (until("]")) { AttrValue e; if (!parse_attr_value(e)) return false; accept(","); }
      a.kind = AttrValue::Unit;
      return ok();
    }
    diag.fail(k.line, "cannot parse attribute value at '" + k.text + "'");
    return false;
  }
  bool parse_attr_dict(std::map<std::string, AttrValue>& d) {
    if (!expect("{")) return false;
    while (until("}")) {
      std::string key = next().text;
      AttrValue v;
      if (accept("=")) { if (!parse_attr_value(v)) return false; }
      else v.kind = AttrValue::Unit;
      d[key] = v;
      accept(",");
    }
    return ok();
  }
  static bool is_scalar_name(const std::string& s) {
    return s == "f64" || s == "f32" || s == "index" || s == "i1" || s == "i32" || s == "i64";
  }
  bool parse_type(Type& ty) {
    const Tok k = next();
    if (k.kind == Tk::MemRef) {
      ty.kind = TypeKind::MemRef;
      std::string s = k.text;
      if (s.compare(0, 2, "T:") == 0) { ty.tensor = true; s = s.substr(2); }
      size_t pos = 0;
      std::vector<std::string> parts;
      while (true) {
        size_t x = s.find('x', pos);
        // the element type itself may not contain 'x' before its end (f64, f32, index: 'index' has an x!)
        std::string piece = s.substr(pos, x == std::string::npos ? std::string::npos : x - pos);
        bool dim = !piece.empty() && (piece == "?" || std::isdigit((unsigned char)piece[0]));
        if (x == std::string::npos || !dim) { parts.push_back(s.substr(pos)); break; }
        parts.push_back(piece);
        pos = x + 1;
      }
      for (size_t i = 0; i + 1 < parts.size(); ++i) ty.shape.push_back(parts[i] == "?" ? -1 : std::strtoll(parts[i].c_str(), nullptr, 10));
      ty.elem = parts.back();
      while (!ty.elem.empty() && std::isspace((unsigned char)ty.elem.back())) ty.elem.pop_back();
      if (!is_scalar_name(ty.elem)) { diag.fail(k.line, "unsupported memref element type '" + ty.elem + "'"); return false; }
      return true;
    }
    if (k.text == "!neptune_ir.temp" || k.text == "!neptune_ir.field") {
      // NeptuneIRTypes.td:22-33 / 47-58: `<` struct(params) `>`
      ty.kind = k.text == "!neptune_ir.temp" ? TypeKind::Temp : TypeKind::Field;
      if (!expect("<")) return false;
      bool has_b = false;
      while (until(">")) {
        std::string key = next().text;
        if (!expect("=")) return false;
        if (key == "element") ty.elem = next().text;
        else if (key == "bounds") { AttrValue a; if (!parse_attr_value(a)) return false; if (a.kind != AttrValue::BoundsK) { diag.fail(k.line, "bounds parameter must be a #neptune_ir.bounds"); return false; } ty.bounds = a.bounds; has_b = true; }
        else if (key == "location") { AttrValue a; if (!parse_attr_value(a)) return false; ty.location = a.s; }
        else { diag.fail(k.line, "unknown type parameter '" + key + "'"); return false; }
        accept(",");
      }
      if (!has_b || ty.elem.empty()) { diag.fail(k.line, "temp/field type needs element and bounds"); return false; }
      return ok();
    }
    if (k.kind == Tk::Id && k.text[0] == '!') {
      auto it = type_alias.find(k.text);
      if (it == type_alias.end()) { diag.fail(k.line, "unknown type alias " + k.text); return false; }
      ty = it->second;
      return true;
    }
    if (is_scalar_name(k.text)) { ty.kind = TypeKind::Scalar; ty.elem = k.text; return true; }
    diag.fail(k.line, "cannot parse type at '" + k.text + "'");
    return false;
  }
  bool parse_type_list_parens(std::vector<Type>& v) {
    if (!expect("(")) return false;
    while (until(")")) { Type ty; if (!parse_type(ty)) return false; v.push_back(ty); accept(","); }
    return ok();
  }
  bool parse_result_types(std::vector<Type>& v) {
    if (is("(")) return parse_type_list_parens(v);
    Type ty;
    if (!parse_type(ty)) return false;
    v.push_back(ty);
    return true;
  }
  void parse_operands(std::vector<std::string>& v) {
    while (is_value()) { v.push_back(next().text); if (!accept(",")) break; }
  }

  bool parse_block_label(Block& b) {  // optional '^bb0(%a: T, ...):'
    if (peek().kind == Tk::Id && peek().text[0] == '^') {
      next();
      if (accept("(")) {
        while (until(")")) {
          BlockArg a;
          a.name = next().text;
          if (!expect(":") || !parse_type(a.type)) return false;
          b.args.push_back(a);
          accept(",");
        }
      }
      if (!expect(":")) return false;
    }
    return ok();
  }
  bool parse_ops_until_close(Block& b) {
    while (until("}")) {
      if (peek().kind == Tk::Eof) { diag.fail(peek().line, "unexpected end of input inside a region"); return false; }
      auto op = std::make_unique<Op>();
      if (!parse_op(*op)) return false;
      b.ops.push_back(std::move(op));
    }
    return ok();
  }
  // skip the text of an op outside the hot-path subset; the SSA values it mentions are kept as its operands, so that
  // the emitter can tell which stencil results a solver op consumes (emit_hip.cpp: outlining)
  void skip_opaque(Op* op = nullptr) {
    int depth = 0;
    while (true) {
      const Tok& k = peek();
      if (k.kind == Tk::Eof) return;
      if (depth == 0) {
        if (k.kind == Tk::Punct && k.text == "}") return;
        if (k.kind == Tk::Id && std::strchr("%@^#!", k.text[0]) == nullptr &&
            (k.text.find('.') != std::string::npos || k.text == "return")) return;
        if (is_value()) {
          size_t j = 1;
          while (is(",", j) && is_value(j + 1)) j += 2;
          if (is("=", j)) return;
        }
      }
      if (k.kind == Tk::Punct && std::strchr("{([<", k.text[0])) ++depth;
      else if (k.kind == Tk::Punct && std::strchr("})]>", k.text[0])) --depth;
      if (op && k.kind == Tk::Id && k.text[0] == '%') op->operands.push_back(k.text);
      next();
    }
  }

  bool parse_op(Op& op) {
    op.line = peek().line;
    if (is_value()) {
      while (true) { op.results.push_back(next().text); if (!accept(",")) break; }
      if (!expect("=")) return false;
    }
    op.name = next().text;
    const std::string& n = op.name;
    if (n == "neptune_ir.apply") {
      // `(` $inputs `)` attr-dict-with-keyword `:` functional-type($inputs, $result) $body   (NeptuneIROps.td:190-194)
      if (!expect("(")) return false;
      parse_operands(op.operands);
      if (!expect(")")) return false;
      if (accept("attributes")) { if (!parse_attr_dict(op.attrs)) return false; }
      if (!expect(":")) return false;
      if (!parse_type_list_parens(op.types)) return false;  // input types first ...
      if (!expect("->")) return false;
      Type res;
      if (!parse_type(res)) return false;
      op.types.push_back(res);                                // ... result type last
      if (!expect("{")) return false;
      auto blk = std::make_unique<Block>();
      if (!parse_block_label(*blk) || !parse_ops_until_close(*blk)) return false;
      op.regions.push_back(std::move(blk));
      return true;
    }
    if (n == "neptune_ir.access") {
      // $input $offsets attr-dict `:` type($input) `->` type($result)   (NeptuneIROps.td:222-225)
      op.operands.push_back(next().text);
      if (!parse_int_list(op.offsets)) return false;
      if (is("{")) { if (!parse_attr_dict(op.attrs)) return false; }
      Type a, b;
      if (!expect(":") || !parse_type(a) || !expect("->") || !parse_type(b)) return false;
      op.types = {a, b};
      return true;
    }
    if (n == "neptune_ir.wrap" || n == "neptune_ir.unwrap" || n == "neptune_ir.load" || n == "neptune_ir.as_tensor" ||
        n == "neptune_ir.from_tensor") {
      // $x attr-dict `:` type($x) `->` type($result)   (NeptuneIROps.td:31-33, 55-57, 79-81, 553-556, 588-591)
      op.operands.push_back(next().text);
      if (is("{")) { if (!parse_attr_dict(op.attrs)) return false; }
      Type a, b;
      if (!expect(":") || !parse_type(a) || !expect("->") || !parse_type(b)) return false;
      op.types = {a, b};
      return true;
    }
    if (n == "neptune_ir.store") {
      // $value `to` $var_field attr-dict `:` type($value) `to` type($var_field)   (NeptuneIROps.td:253-256)
      op.operands.push_back(next().text);
      if (!expect("to")) return false;
      op.operands.push_back(next().text);
      if (is("{")) { if (!parse_attr_dict(op.attrs)) return false; }
      Type a, b;
      if (!expect(":") || !parse_type(a) || !expect("to") || !parse_type(b)) return false;
      op.types = {a, b};
      return true;
    }
    if (n == "neptune_ir.time_advance") {
      // $state `,` $dt attr-dict `:` type($state) `,` type($dt) `->` type($result)   (NeptuneIROps.td:766-770)
      parse_operands(op.operands);
      if (is("{")) { if (!parse_attr_dict(op.attrs)) return false; }
      Type a, b, c;
      if (!expect(":") || !parse_type(a) || !expect(",") || !parse_type(b) || !expect("->") || !parse_type(c)) return false;
      op.types = {a, b, c};
      // only the explicit method (TimeMethod 0, NeptuneIRAttrs.td:78-85) with an rhs operator is part of
      // the stencil path; implicit / runtime methods go to the solver runtime on the host
      auto m = op.attrs.find("method");
      auto r = op.attrs.find("rhs");
      const bool is_explicit = m != op.attrs.end() && m->second.kind == AttrValue::Int && m->second.i == 0 &&
                               r != op.attrs.end() && r->second.kind == AttrValue::Symbol;
      if (is_explicit) op.callee = r->second.s;
      else op.opaque = true;
      return true;
    }
    if (n == "neptune_ir.reduce") {
      // $input (`in` $bounds^)? attr-dict `:` type($input) `->` type($result)   (NeptuneIROps.td:293-296)
      op.operands.push_back(next().text);
      if (accept("in")) { AttrValue b; if (!parse_attr_value(b)) return false; op.attrs["bounds"] = b; }
      if (is("{")) { if (!parse_attr_dict(op.attrs)) return false; }
      Type a, b;
      if (!expect(":") || !parse_type(a) || !expect("->") || !parse_type(b)) return false;
      op.types = {a, b};
      return true;
    }
    if (n == "neptune_ir.apply_linear" || n == "neptune_ir.apply_nonlinear") {
      // $op `(` $inputs `)` attr-dict `:` functional-type($inputs, $results)   (NeptuneIROps.td:482-485)
      op.callee = next().text.substr(1);
      if (!expect("(")) return false;
      parse_operands(op.operands);
      if (!expect(")")) return false;
      if (accept("attributes")) { if (!parse_attr_dict(op.attrs)) return false; }
      else if (is("{")) { if (!parse_attr_dict(op.attrs)) return false; }
      if (!expect(":")) return false;
      if (!parse_type_list_parens(op.types)) return false;
      if (!expect("->")) return false;
      std::vector<Type> res;
      if (!parse_result_types(res)) return false;
      for (auto& r : res) op.types.push_back(r);
      return true;
    }
    if (n == "neptune_ir.yield" || n == "neptune_ir.return" || n == "func.return" || n == "return" || n == "scf.yield") {
      parse_operands(op.operands);
      if (!op.operands.empty()) {
        if (!expect(":")) return false;
        for (size_t i = 0; i < op.operands.size(); ++i) { Type ty; if (!parse_type(ty)) return false; op.types.push_back(ty); accept(","); }
      }
      return true;
    }
    if (n == "arith.constant") {
      const Tok v = next();
      Type ty;
      if (v.text == "true" || v.text == "false") {
        op.literal = v.text;
        ty.kind = TypeKind::Scalar;
        ty.elem = "i1";
        if (accept(":")) { if (!parse_type(ty)) return false; }
      } else {
        if (v.kind != Tk::Num) { diag.fail(v.line, "unsupported arith.constant value '" + v.text + "'"); return false; }
        op.literal = v.text;
        if (!expect(":") || !parse_type(ty)) return false;
      }
      op.types = {ty};
      return true;
    }
    if (n == "arith.cmpi" || n == "arith.cmpf") {
      op.predicate = next().text;
      if (!expect(",")) return false;
      parse_operands(op.operands);
      Type ty;
      if (!expect(":") || !parse_type(ty)) return false;
      op.types = {ty};
      return true;
    }
    if (n == "arith.select") {
      parse_operands(op.operands);
      Type ty;
      if (!expect(":") || !parse_type(ty)) return false;
      if (accept(",")) { if (!parse_type(ty)) return false; }
      op.types = {ty};
      return true;
    }
    if (n == "scf.if") {
      op.operands.push_back(next().text);
      if (accept("->")) { if (!parse_result_types(op.types)) return false; }
      if (!expect("{")) return false;
      auto th = std::make_unique<Block>();
      if (!parse_ops_until_close(*th)) return false;
      op.regions.push_back(std::move(th));
      if (accept("else")) {
        if (!expect("{")) return false;
        auto el = std::make_unique<Block>();
        if (!parse_ops_until_close(*el)) return false;
        op.regions.push_back(std::move(el));
      }
      return true;
    }
    if (n.compare(0, 6, "arith.") == 0 || n.compare(0, 5, "math.") == 0) {
      parse_operands(op.operands);
      if (is("{")) { if (!parse_attr_dict(op.attrs)) return false; }
      Type a;
      if (!expect(":") || !parse_type(a)) return false;
      op.types = {a};
      if (accept("to")) { Type b; if (!parse_type(b)) return false; op.types.push_back(b); }
      return true;
    }
    if (n.compare(0, 11, "neptune_ir.") == 0) {
      // solver / time-stepping surface: outside the stencil hot path (stays on the host path)
      op.opaque = true;
      skip_opaque(&op);
      return true;
    }
    diag.fail(op.line, "unsupported operation '" + n + "'");
    return false;
  }

  bool parse_function(Module& m) {
    const Tok head = next();
    if (head.text != "func.func" && head.text != "neptune_ir.linear_opdef" && head.text != "neptune_ir.nonlinear_opdef") {
      diag.fail(head.line, "expected func.func or a neptune_ir opdef at module level, got '" + head.text + "'");
      return false;
    }
    auto f = std::make_unique<Function>();
    f->line = head.line;
    if (head.text == "func.func") {
      while (is("private") || is("public")) next();
      f->name = next().text.substr(1);
      f->kind = FuncKind::Func;
      if (!expect("(")) return false;
      while (until(")")) {
        BlockArg a;
        a.name = next().text;
        if (!expect(":") || !parse_type(a.type)) return false;
        f->body.args.push_back(a);
        f->arg_types.push_back(a.type);
        accept(",");
      }
      if (accept("->")) { if (!parse_result_types(f->result_types)) return false; }
      if (accept("attributes")) { std::map<std::string, AttrValue> d; if (!parse_attr_dict(d)) return false; }
      if (!expect("{")) return false;
      if (!parse_ops_until_close(f->body)) return false;
    } else {
      // $sym_name attr-dict-with-keyword `:` $function_type $body   (NeptuneIROps.td:347-349, 414-416)
      f->kind = head.text == "neptune_ir.linear_opdef" ? FuncKind::LinearOpDef : FuncKind::NonlinearOpDef;
      f->name = next().text.substr(1);
      if (accept("attributes")) { std::map<std::string, AttrValue> d; if (!parse_attr_dict(d)) return false; }
      if (!expect(":")) return false;
      if (!parse_type_list_parens(f->arg_types)) return false;
      if (!expect("->")) return false;
      if (!parse_result_types(f->result_types)) return false;
      if (accept("attributes")) { std::map<std::string, AttrValue> d; if (!parse_attr_dict(d)) return false; }
      if (!expect("{")) return false;
      if (!parse_block_label(f->body) || !parse_ops_until_close(f->body)) return false;
    }
    if (m.find(f->name)) { diag.fail(f->line, "redefinition of symbol @" + f->name); return false; }
    m.funcs.push_back(std::move(f));
    return ok();
  }

  bool parse_top(Module& m) {
    while (ok() && peek().kind != Tk::Eof) {
      const Tok& k = peek();
      if (k.kind == Tk::Id && k.text[0] == '#' && is("=", 1)) {
        std::string name = next().text;
        next();
        AttrValue a;
        if (!parse_attr_value(a)) return false;
        attr_alias[name] = a;
      } else if (k.kind == Tk::Id && k.text[0] == '!' && is("=", 1)) {
        std::string name = next().text;
        next();
        Type ty;
        if (!parse_type(ty)) return false;
        type_alias[name] = ty;
      } else if (k.text == "module") {
        next();
        if (peek().kind == Tk::Id && peek().text[0] == '@') next();
        if (accept("attributes")) { std::map<std::string, AttrValue> d; if (!parse_attr_dict(d)) return false; }
        if (!expect("{")) return false;
        while (until("}")) {
          if (peek().kind == Tk::Eof) { diag.fail(peek().line, "unexpected end of input inside module"); return false; }
          // stray module-level solver ops (the reference's Python builder can leave e.g. an
          // assemble_matrix there): nothing to lower, skip them
          const bool stray_result = is_value();
          const bool stray_op = peek().kind == Tk::Id && peek().text.compare(0, 11, "neptune_ir.") == 0 &&
                                peek().text != "neptune_ir.linear_opdef" && peek().text != "neptune_ir.nonlinear_opdef";
          if (stray_result || stray_op) {
            next();
            skip_opaque();
            continue;
          }
          if (!parse_function(m)) return false;
        }
      } else if (k.text == "func.func" || k.text == "neptune_ir.linear_opdef" || k.text == "neptune_ir.nonlinear_opdef") {
        if (!parse_function(m)) return false;
      } else {
        diag.fail(k.line, "unexpected top-level token '" + k.text + "'");
        return false;
      }
    }
    return ok();
  }
};

}  // namespace

bool parse_module(const std::string& text, Module& out, Diag& diag) {
  Parser ps(diag);
  if (!tokenize(text, ps.t, diag)) return false;
  return ps.parse_top(out) && diag.ok;
}

}  // namespace neptune_lowering

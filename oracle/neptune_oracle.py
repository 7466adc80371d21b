"""CPU oracle for NeptuneIR's stencil hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The product path (neptune-pde-solver_amd/) never does, and fails loudly without its HIP
library.

PARITY UNPINNED: the reference (levia-than/neptune-pde-solver) ships no expected outputs for
this path (its smoke scripts only print, test/smoke_tests/smoke_apply.sh:76-80; its lit tests
only CHECK symbol names, test/mlir_tests/solver-pipeline-to-llvm.mlir:49-55) and it cannot be
built here (C++ against MLIR/LLVM 21.x, third_party/llvm-project is an empty submodule).  This
oracle is therefore a restatement pinned only by known-answer vectors hand-derived from the
reference's own inputs (tests/golden/kat_*.json, derivations in tests/golden/README.md).

What is restated, and from where (all paths relative to the reference root):
  * field/temp -> dense row-major buffer of shape ub-lb   lib/Passes/DataflowLowering.cpp:41-49
  * wrap / unwrap / load are aliases (no copy)             :131-159
  * apply: fresh result, copy-through of input 0, loop nest over apply.bounds in row-major
    order, access = in[p + off - in_lb] with NO bounds check, yield stored at p - out_lb
                                                           :258-448
  * store: whole-buffer copy, or logical sub-box copy where each side uses its own origin
                                                           :165-220
  * linear_opdef / nonlinear_opdef = function, apply_linear / apply_nonlinear = call
                                                           lib/Passes/StructureLowering.cpp:30-124
  * region signature (rank x index, then one temp per input), single yield
                                                           lib/Dialect/NeptuneIR/NeptuneIRVerifier.cpp:141-171,
                                                           lib/Passes/VerifyAndAnnotate.cpp:87-214
  * arithmetic: each arith op evaluated on its own, in textual order, strict IEEE, no FMA, no
    reassociation (the pipeline has no fusing/vectorising pass,
    lib/Pipeline/NeptuneIRPassesPipeline.cpp:9-46)

Evaluation strategy: iterations of an apply are independent (the body reads only inputs, never
the result), so instead of the reference's scalar loops every op is applied to whole numpy
arrays over apply.bounds.  numpy evaluates each ufunc separately in IEEE binary64/binary32, so
the bits equal those of the scalar loop.  `scf.if` is evaluated with real control flow
semantics: each branch only on the points that take it (the reference's else-branch may read
out of bounds on points that never execute it, smoke_time_advance_nonlinear.mlir:29-70).

Deliberate difference from the reference: an access that leaves its input's buffer is undefined
behaviour there (test/smoke_tests/smoke_apply.mlir:4-9 does it); here it raises OutOfBounds.

This parser is independent of the product's C++ parser (csrc/lowering): two implementations
that must agree on every fixture.
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np


_ELEMENTARY = {"math.exp": np.exp, "math.log": np.log, "math.sin": np.sin, "math.cos": np.cos, "math.tanh": np.tanh,
               "math.powf": np.power}


class OracleError(Exception):
    pass


class OutOfBounds(OracleError):
    pass


class Unsupported(OracleError):
    pass


# --------------------------------------------------------------------------------------
# types
# --------------------------------------------------------------------------------------
@dataclass(frozen=True)
class Bounds:
    lb: Tuple[int, ...]
    ub: Tuple[int, ...]

    @property
    def rank(self) -> int:
        return len(self.lb)

    @property
    def shape(self) -> Tuple[int, ...]:
        return tuple(u - l for l, u in zip(self.lb, self.ub))


@dataclass(frozen=True)
class TempType:  # also used for field types; `kind` tells them apart
    kind: str  # "temp" | "field"
    element: str
    bounds: Bounds
    location: str


@dataclass(frozen=True)
class MemRefType:
    shape: Tuple[Optional[int], ...]
    element: str


@dataclass(frozen=True)
class ScalarType:
    name: str  # f64 f32 index i1 i32 i64


_NP = {"f64": np.float64, "f32": np.float32, "index": np.int64, "i64": np.int64, "i32": np.int32,
       "i1": np.bool_}


def np_dtype(name: str):
    if name not in _NP:
        raise Unsupported(f"element type {name}")
    return _NP[name]


# --------------------------------------------------------------------------------------
# tokenizer
# --------------------------------------------------------------------------------------
_TOKEN_RE = re.compile(
    r"""
    (?P<ws>\s+|//[^\n]*)
  | (?P<arrow>->)
  | (?P<str>"(?:[^"\\]|\\.)*")
  | (?P<num>[-+]?(?:0x[0-9a-fA-F]+|(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?))
  | (?P<id>[%@^\#!]?[A-Za-z_][A-Za-z0-9_.$]*|[%@^\#!]\d+)
  | (?P<punct>[{}()\[\]<>,:=?*])
    """,
    re.VERBOSE,
)


@dataclass
class Tok:
    kind: str
    text: str
    pos: int


def tokenize(src: str) -> List[Tok]:
    toks: List[Tok] = []
    i = 0
    n = len(src)
    while i < n:
        m = _TOKEN_RE.match(src, i)
        if not m:
            raise OracleError(f"cannot tokenize at offset {i}: {src[i:i+30]!r}")
        kind = m.lastgroup
        text = m.group()
        if kind == "id" and text in ("memref", "tensor"):   # tensor<...>: as_tensor / from_tensor casts (:705-733)
            # memref<?x?xf64> : take the bracketed text raw ('?x?xf64' is not tokenizable)
            j = m.end()
            while j < n and src[j].isspace():
                j += 1
            if j < n and src[j] == "<":
                depth = 0
                k = j
                while k < n:
                    if src[k] == "<":
                        depth += 1
                    elif src[k] == ">":
                        depth -= 1
                        if depth == 0:
                            break
                    k += 1
                toks.append(Tok("memref", src[j + 1:k], i))
                i = k + 1
                continue
        if kind != "ws":
            toks.append(Tok(kind, text, i))
        i = m.end()
    toks.append(Tok("eof", "", n))
    return toks


# --------------------------------------------------------------------------------------
# IR
# --------------------------------------------------------------------------------------
@dataclass
class Op:
    name: str
    results: List[str]
    operands: List[str]
    attrs: Dict[str, object] = field(default_factory=dict)
    types: List[object] = field(default_factory=list)
    regions: List["Block"] = field(default_factory=list)


@dataclass
class Block:
    args: List[Tuple[str, object]]
    ops: List[Op]


@dataclass
class Function:
    name: str
    kind: str  # "func" | "linear_opdef" | "nonlinear_opdef"
    arg_types: List[object]
    result_types: List[object]
    body: Block


class Parser:
    def __init__(self, src: str):
        self.toks = tokenize(src)
        self.p = 0
        self.attr_alias: Dict[str, object] = {}
        self.type_alias: Dict[str, object] = {}

    # -- token helpers
    def peek(self, k: int = 0) -> Tok:
        return self.toks[min(self.p + k, len(self.toks) - 1)]

    def next(self) -> Tok:
        t = self.toks[self.p]
        self.p += 1
        return t

    def accept(self, text: str) -> bool:
        if self.peek().text == text and self.peek().kind != "str":
            self.p += 1
            return True
        return False

    def expect(self, text: str) -> Tok:
        t = self.next()
        if t.text != text:
            raise OracleError(f"expected {text!r}, got {t.text!r} at offset {t.pos}")
        return t

    # -- attributes and types
    def parse_int_list(self) -> Tuple[int, ...]:
        self.expect("[")
        vals: List[int] = []
        while not self.accept("]"):
            t = self.next()
            if t.kind != "num":
                raise OracleError(f"expected integer, got {t.text!r}")
            vals.append(int(t.text, 0))
            self.accept(",")
        return tuple(vals)

    def parse_bounds_attr(self) -> Bounds:
        # after '#neptune_ir.bounds'
        self.expect("<")
        lb = ub = None
        while not self.accept(">"):
            key = self.next().text
            self.expect("=")
            vals = self.parse_int_list()
            if key == "lb":
                lb = vals
            elif key == "ub":
                ub = vals
            else:
                raise OracleError(f"unknown bounds key {key}")
            self.accept(",")
        if lb is None or ub is None or len(lb) != len(ub):
            raise OracleError("bounds needs lb and ub of equal rank")
        return Bounds(lb, ub)

    def parse_attr_value(self):
        t = self.peek()
        if t.text == "#neptune_ir.bounds":
            self.next()
            return self.parse_bounds_attr()
        if t.text == "#neptune_ir.location":
            self.next()
            self.expect("<")
            s = self.next().text.strip('"')
            self.expect(">")
            return ("location", s)
        if t.kind == "id" and t.text.startswith("#"):
            self.next()
            if t.text not in self.attr_alias:
                raise OracleError(f"unknown attribute alias {t.text}")
            return self.attr_alias[t.text]
        if t.kind == "id" and t.text.startswith("@"):
            self.next()
            return ("symbol", t.text[1:])
        if t.kind == "str":
            self.next()
            return t.text.strip('"')
        if t.kind == "num":
            self.next()
            val: object = float(t.text) if re.search(r"[.eE]", t.text) and not t.text.startswith("0x") else int(t.text, 0)
            if self.accept(":"):
                self.parse_type()
            return val
        if t.text == "[":
            # generic array attribute
            self.next()
            vals = []
            while not self.accept("]"):
                vals.append(self.parse_attr_value())
                self.accept(",")
            return vals
        if t.text in ("true", "false"):
            self.next()
            return t.text == "true"
        raise OracleError(f"cannot parse attribute value at {t.text!r}")

    def parse_attr_dict(self) -> Dict[str, object]:
        d: Dict[str, object] = {}
        self.expect("{")
        while not self.accept("}"):
            key = self.next().text
            if self.accept("="):
                d[key] = self.parse_attr_value()
            else:
                d[key] = True
            self.accept(",")
        return d

    def parse_type(self):
        t = self.next()
        if t.kind == "memref":
            parts = t.text.strip().split("x")
            elem = parts[-1].strip()
            shape = tuple(None if s.strip() == "?" else int(s) for s in parts[:-1])
            return MemRefType(shape, elem)
        if t.text in ("!neptune_ir.temp", "!neptune_ir.field"):
            kind = t.text.split(".")[1]
            self.expect("<")
            element = bounds = None
            location = ""
            while not self.accept(">"):
                key = self.next().text
                self.expect("=")
                if key == "element":
                    element = self.next().text
                elif key == "bounds":
                    bounds = self.parse_attr_value()
                elif key == "location":
                    loc = self.parse_attr_value()
                    location = loc[1] if isinstance(loc, tuple) else str(loc)
                else:
                    raise OracleError(f"unknown type parameter {key}")
                self.accept(",")
            if element is None or not isinstance(bounds, Bounds):
                raise OracleError("temp/field type needs element and bounds")
            return TempType(kind, element, bounds, location)
        if t.kind == "id" and t.text.startswith("!"):
            if t.text not in self.type_alias:
                raise OracleError(f"unknown type alias {t.text}")
            return self.type_alias[t.text]
        if t.text in _NP:
            return ScalarType(t.text)
        raise OracleError(f"cannot parse type at {t.text!r} (offset {t.pos})")

    def parse_type_list_parens(self) -> List[object]:
        tys: List[object] = []
        self.expect("(")
        while not self.accept(")"):
            tys.append(self.parse_type())
            self.accept(",")
        return tys

    def parse_result_types(self) -> List[object]:
        if self.peek().text == "(":
            return self.parse_type_list_parens()
        return [self.parse_type()]

    # -- module level
    def parse_module(self) -> "Module":
        funcs: Dict[str, Function] = {}
        while self.peek().kind != "eof":
            t = self.peek()
            if t.kind == "id" and t.text.startswith("#") and self.peek(1).text == "=":
                self.next()
                self.next()
                self.attr_alias[t.text] = self.parse_attr_value()
            elif t.kind == "id" and t.text.startswith("!") and self.peek(1).text == "=":
                self.next()
                self.next()
                self.type_alias[t.text] = self.parse_type()
            elif t.text == "module":
                self.next()
                if self.peek().kind == "id" and self.peek().text.startswith("@"):
                    self.next()
                if self.peek().text == "attributes":
                    self.next()
                    self.parse_attr_dict()
                self.expect("{")
                while not self.accept("}"):
                    t2 = self.peek()
                    stray = t2.text.startswith("%") or (t2.text.startswith("neptune_ir.") and
                                                         t2.text not in ("neptune_ir.linear_opdef", "neptune_ir.nonlinear_opdef"))
                    if stray:   # module-level solver op left by the Python builder: nothing to evaluate
                        self.next()
                        self._skip_opaque_op()
                        continue
                    f = self.parse_function()
                    funcs[f.name] = f
            elif t.text in ("func.func", "neptune_ir.linear_opdef", "neptune_ir.nonlinear_opdef"):
                f = self.parse_function()
                funcs[f.name] = f
            else:
                raise OracleError(f"unexpected top-level token {t.text!r} at offset {t.pos}")
        return Module(funcs)

    def parse_function(self) -> Function:
        t = self.next()
        if t.text == "func.func":
            while self.peek().text in ("private", "public"):
                self.next()
            name = self.next().text[1:]
            self.expect("(")
            args: List[Tuple[str, object]] = []
            while not self.accept(")"):
                an = self.next().text
                self.expect(":")
                args.append((an, self.parse_type()))
                self.accept(",")
            results: List[object] = []
            if self.accept("->"):
                results = self.parse_result_types()
            if self.peek().text == "attributes":
                self.next()
                self.parse_attr_dict()
            self.expect("{")
            ops = self.parse_ops_until_close()
            return Function(name, "func", [a[1] for a in args], results, Block(args, ops))
        if t.text in ("neptune_ir.linear_opdef", "neptune_ir.nonlinear_opdef"):
            kind = t.text.split(".")[1]
            name = self.next().text[1:]
            if self.peek().text == "attributes":
                self.next()
                self.parse_attr_dict()
            self.expect(":")
            arg_types = self.parse_type_list_parens()
            self.expect("->")
            results = self.parse_result_types()
            if self.peek().text == "attributes":
                self.next()
                self.parse_attr_dict()
            self.expect("{")
            blk = self.parse_block_with_label()
            return Function(name, kind, arg_types, results, blk)
        raise OracleError(f"expected a function-like op, got {t.text!r} at offset {t.pos}")

    def parse_block_with_label(self) -> Block:
        """'^bb0(%a: T, ...):' ops '}'  (the opening '{' is already consumed)"""
        args: List[Tuple[str, object]] = []
        if self.peek().kind == "id" and self.peek().text.startswith("^"):
            self.next()
            if self.accept("("):
                while not self.accept(")"):
                    an = self.next().text
                    self.expect(":")
                    args.append((an, self.parse_type()))
                    self.accept(",")
            self.expect(":")
        ops = self.parse_ops_until_close()
        return Block(args, ops)

    def parse_ops_until_close(self) -> List[Op]:
        ops: List[Op] = []
        while not self.accept("}"):
            ops.append(self.parse_op())
        return ops

    # -- operations
    def parse_operand_list(self, close: Optional[str] = None) -> List[str]:
        vals: List[str] = []
        while self.peek().kind == "id" and self.peek().text.startswith("%"):
            vals.append(self.next().text)
            if not self.accept(","):
                break
        return vals

    def parse_op(self) -> Op:
        results: List[str] = []
        if self.peek().kind == "id" and self.peek().text.startswith("%"):
            while True:
                results.append(self.next().text)
                if not self.accept(","):
                    break
            self.expect("=")
        t = self.next()
        name = t.text
        op = Op(name, results, [])
        if name == "neptune_ir.apply":
            self.expect("(")
            op.operands = self.parse_operand_list()
            self.expect(")")
            if self.peek().text == "attributes":
                self.next()
                op.attrs = self.parse_attr_dict()
            elif self.peek().text == "{" and self.peek(1).kind == "id" and self.peek(2).text == "=":
                op.attrs = self.parse_attr_dict()
            self.expect(":")
            op.types = [self.parse_type_list_parens()]
            self.expect("->")
            op.types.append(self.parse_type())
            self.expect("{")
            op.regions = [self.parse_block_with_label()]
            return op
        if name == "neptune_ir.access":
            op.operands = [self.next().text]
            op.attrs["offsets"] = self.parse_int_list()
            if self.peek().text == "{":
                op.attrs.update(self.parse_attr_dict())
            self.expect(":")
            op.types = [self.parse_type()]
            self.expect("->")
            op.types.append(self.parse_type())
            return op
        if name in ("neptune_ir.wrap", "neptune_ir.unwrap", "neptune_ir.load", "neptune_ir.as_tensor", "neptune_ir.from_tensor"):
            op.operands = [self.next().text]
            if self.peek().text == "{":
                op.attrs = self.parse_attr_dict()
            self.expect(":")
            op.types = [self.parse_type()]
            self.expect("->")
            op.types.append(self.parse_type())
            return op
        if name == "neptune_ir.store":
            op.operands = [self.next().text]
            self.expect("to")
            op.operands.append(self.next().text)
            if self.peek().text == "{":
                op.attrs = self.parse_attr_dict()
            self.expect(":")
            op.types = [self.parse_type()]
            self.expect("to")
            op.types.append(self.parse_type())
            return op
        if name == "neptune_ir.time_advance":
            # $state `,` $dt attr-dict `:` type($state) `,` type($dt) `->` type($result)   (NeptuneIROps.td:766-770)
            op.operands = self.parse_operand_list()
            if self.peek().text == "{":
                op.attrs.update(self.parse_attr_dict())
            self.expect(":")
            op.types = [self.parse_type()]
            self.expect(",")
            op.types.append(self.parse_type())
            self.expect("->")
            op.types.append(self.parse_type())
            explicit = op.attrs.get("method") == 0 and isinstance(op.attrs.get("rhs"), tuple)
            if not explicit:
                op.attrs["opaque"] = True      # implicit / runtime methods: solver surface
            return op
        if name == "neptune_ir.reduce":
            # $input (`in` $bounds^)? attr-dict `:` type($input) `->` type($result)   (NeptuneIROps.td:293-296)
            op.operands = [self.next().text]
            if self.accept("in"):
                op.attrs["bounds"] = self.parse_attr_value()
            if self.peek().text == "{":
                op.attrs.update(self.parse_attr_dict())
            self.expect(":")
            op.types = [self.parse_type()]
            self.expect("->")
            op.types.append(self.parse_type())
            return op
        if name in ("neptune_ir.apply_linear", "neptune_ir.apply_nonlinear"):
            op.attrs["callee"] = self.next().text[1:]
            self.expect("(")
            op.operands = self.parse_operand_list()
            self.expect(")")
            if self.peek().text == "attributes":
                self.next()
                op.attrs.update(self.parse_attr_dict())
            elif self.peek().text == "{":
                op.attrs.update(self.parse_attr_dict())
            self.expect(":")
            op.types = [self.parse_type_list_parens()]
            self.expect("->")
            op.types.append(self.parse_result_types())
            return op
        if name in ("neptune_ir.yield", "neptune_ir.return", "func.return", "return", "scf.yield"):
            op.operands = self.parse_operand_list()
            if op.operands:
                self.expect(":")
                for _ in op.operands:
                    op.types.append(self.parse_type())
                    self.accept(",")
            return op
        if name == "arith.constant":
            tk = self.next()
            if tk.text in ("true", "false"):
                op.attrs["value"] = tk.text == "true"
                op.types = [ScalarType("i1")]
                if self.accept(":"):
                    op.types = [self.parse_type()]
                return op
            if tk.kind != "num":
                raise Unsupported(f"arith.constant value {tk.text!r}")
            self.expect(":")
            ty = self.parse_type()
            op.types = [ty]
            op.attrs["literal"] = tk.text
            return op
        if name in ("arith.cmpi", "arith.cmpf"):
            op.attrs["predicate"] = self.next().text
            self.expect(",")
            op.operands = self.parse_operand_list()
            self.expect(":")
            op.types = [self.parse_type()]
            return op
        if name == "arith.select":
            op.operands = self.parse_operand_list()
            self.expect(":")
            op.types = [self.parse_type()]
            if self.accept(","):
                op.types = [self.parse_type()]
            return op
        if name == "scf.if":
            op.operands = [self.next().text]
            if self.accept("->"):
                op.types = self.parse_result_types()
            self.expect("{")
            op.regions = [Block([], self.parse_ops_until_close())]
            if self.accept("else"):
                self.expect("{")
                op.regions.append(Block([], self.parse_ops_until_close()))
            return op
        if name.startswith("arith.") or name.startswith("math."):
            op.operands = self.parse_operand_list()
            if self.peek().text == "{":
                op.attrs = self.parse_attr_dict()
            self.expect(":")
            op.types = [self.parse_type()]
            if self.accept("to"):
                op.types.append(self.parse_type())
            return op
        if name.startswith("neptune_ir."):
            # solver / time-stepping surface (time_advance, assemble_matrix, solve_*, reduce ...):
            # outside the stencil hot path.  Skip the op's text so the opdefs around it can still
            # be evaluated; running a function that contains it raises Unsupported.
            self._skip_opaque_op()
            op.attrs["opaque"] = True
            return op
        raise Unsupported(f"operation {name!r}")

    def _skip_opaque_op(self) -> None:
        depth = 0
        while True:
            t = self.peek()
            if t.kind == "eof":
                return
            if depth == 0:
                if t.text == "}" and t.kind == "punct":
                    return
                if t.kind == "id" and t.text[0] not in "%@^#!" and ("." in t.text or t.text == "return"):
                    return  # next statement's op name
                if t.kind == "id" and t.text.startswith("%"):
                    k = 1
                    while self.peek(k).text == "," and self.peek(k + 1).text.startswith("%"):
                        k += 2
                    if self.peek(k).text == "=":
                        return  # next statement's result list
            if t.kind == "punct" and t.text in "{([<":
                depth += 1
            elif t.kind == "punct" and t.text in "})]>":
                depth -= 1
            self.next()


# --------------------------------------------------------------------------------------
# evaluation
# --------------------------------------------------------------------------------------
@dataclass
class Buffer:
    """a temp/field/memref value: a numpy array (dense row-major) + its logical origin"""
    data: np.ndarray
    lb: Tuple[int, ...]


class PointSet:
    """The set of logical points an apply body is being evaluated on.

    box mode   : the whole box [lb,ub) -- values are arrays of shape ub-lb, accesses are slices
    gather mode: an explicit list of points (after scf.if narrowed the set) -- values are 1-D
                 arrays, accesses are fancy-indexed gathers
    """

    def __init__(self, lb: Sequence[int], ub: Sequence[int], coords: Optional[List[np.ndarray]] = None):
        self.lb = tuple(int(x) for x in lb)
        self.ub = tuple(int(x) for x in ub)
        self.coords = coords

    @property
    def is_box(self) -> bool:
        return self.coords is None

    @property
    def shape(self) -> Tuple[int, ...]:
        if self.is_box:
            return tuple(u - l for l, u in zip(self.lb, self.ub))
        return (len(self.coords[0]),)

    def index(self, d: int) -> np.ndarray:
        if self.is_box:
            shape = [1] * len(self.lb)
            shape[d] = self.ub[d] - self.lb[d]
            return np.broadcast_to(np.arange(self.lb[d], self.ub[d], dtype=np.int64).reshape(shape), self.shape)
        return self.coords[d]

    def to_gather(self) -> "PointSet":
        if not self.is_box:
            return self
        grids = np.meshgrid(*[np.arange(l, u, dtype=np.int64) for l, u in zip(self.lb, self.ub)], indexing="ij")
        return PointSet(self.lb, self.ub, [g.reshape(-1) for g in grids])

    def subset(self, mask: np.ndarray) -> "PointSet":
        g = self.to_gather()
        m = np.asarray(mask).reshape(-1)
        return PointSet(self.lb, self.ub, [c[m] for c in g.coords])

    def access(self, buf: Buffer, off: Sequence[int]) -> np.ndarray:
        rank = len(self.lb)
        if len(off) != rank or buf.data.ndim != rank:
            raise OracleError("access rank mismatch")
        if self.is_box:
            sl = []
            for d in range(rank):
                lo = self.lb[d] + off[d] - buf.lb[d]  # physical = logical + off - in_lb  (:380-410)
                hi = self.ub[d] + off[d] - buf.lb[d]
                if self.ub[d] > self.lb[d] and (lo < 0 or hi > buf.data.shape[d]):
                    raise OutOfBounds(
                        f"access offset {tuple(off)} leaves the input box along dim {d}: physical [{lo},{hi}) "
                        f"vs extent {buf.data.shape[d]} (undefined behaviour in the reference)")
                sl.append(slice(lo, hi))
            return buf.data[tuple(sl)]
        idx = []
        for d in range(rank):
            c = self.coords[d] + (off[d] - buf.lb[d])
            if c.size and (c.min() < 0 or c.max() >= buf.data.shape[d]):
                raise OutOfBounds(f"access offset {tuple(off)} leaves the input box along dim {d}")
            idx.append(c)
        return buf.data[tuple(idx)]


_CMPI = {
    "eq": np.equal, "ne": np.not_equal, "slt": np.less, "sle": np.less_equal, "sgt": np.greater,
    "sge": np.greater_equal,
}
_CMPF = {
    "oeq": np.equal, "ogt": np.greater, "oge": np.greater_equal, "olt": np.less, "ole": np.less_equal,
    "one": lambda a, b: np.logical_and(np.not_equal(a, b), ~(np.isnan(a) | np.isnan(b))),
    "une": np.not_equal,
}
_BINF = {"arith.addf": np.add, "arith.subf": np.subtract, "arith.mulf": np.multiply, "arith.divf": np.divide}
_BINI = {"arith.addi": np.add, "arith.subi": np.subtract, "arith.muli": np.multiply,
         "arith.andi": np.bitwise_and, "arith.ori": np.bitwise_or, "arith.xori": np.bitwise_xor}


def _const(op: Op):
    ty = op.types[0]
    if "value" in op.attrs:
        return np.bool_(op.attrs["value"])
    lit = op.attrs["literal"]
    dt = np_dtype(ty.name)
    if ty.name in ("f64", "f32"):
        return dt(float(lit)) if not lit.startswith("0x") else dt(np.array(int(lit, 16)).view(dt))
    return dt(int(lit, 0))


class Module:
    def __init__(self, funcs: Dict[str, Function]):
        self.funcs = funcs

    @staticmethod
    def parse(text: str) -> "Module":
        return Parser(text).parse_module()

    # ---- public entry: call a lowered symbol with numpy arrays -------------------------
    def call(self, sym: str, *arrays: np.ndarray) -> Union[np.ndarray, Tuple[np.ndarray, ...], None]:
        """Call `@sym` the way the reference's lowered code would be called.

        opdef @A(x...)      : returns a fresh array (callee-allocated result, caller frees)
        func @entry(out, in): memref arguments are borrowed numpy arrays, mutated in place by
                              neptune_ir.store; the returned array aliases the argument that
                              was unwrapped (same object)."""
        if sym not in self.funcs:
            raise OracleError(f"no symbol @{sym}")
        f = self.funcs[sym]
        if len(arrays) != len(f.arg_types):
            raise OracleError(f"@{sym} takes {len(f.arg_types)} arguments, got {len(arrays)}")
        args: List[Buffer] = []
        for a, ty in zip(arrays, f.arg_types):
            args.append(self._bind_arg(a, ty))
        res = self._run_function(f, args)
        outs = tuple(r.data if isinstance(r, Buffer) else r for r in res)
        if not outs:
            return None
        return outs[0] if len(outs) == 1 else outs

    def _bind_arg(self, a: np.ndarray, ty) -> Buffer:
        if isinstance(ty, TempType):
            if tuple(a.shape) != ty.bounds.shape:
                raise OracleError(f"argument shape {a.shape} != {ty.bounds.shape}")
            if a.dtype != np_dtype(ty.element):
                raise OracleError(f"argument dtype {a.dtype} != {ty.element}")
            if not a.flags["C_CONTIGUOUS"]:
                raise OracleError("temps are dense row-major buffers")
            return Buffer(a, ty.bounds.lb)
        if isinstance(ty, MemRefType):
            if a.ndim != len(ty.shape) or a.dtype != np_dtype(ty.element):
                raise OracleError("memref argument rank/dtype mismatch")
            return Buffer(a, tuple([0] * a.ndim))
        raise Unsupported(f"argument type {ty}")

    # ---- function bodies (host-level ops) ------------------------------------------------
    def _run_function(self, f: Function, args: List[Buffer]) -> List[Buffer]:
        env: Dict[str, object] = {}
        for (name, _), val in zip(f.body.args, args):
            env[name] = val
        for op in f.body.ops:
            n = op.name
            if n in ("neptune_ir.wrap", "neptune_ir.load", "neptune_ir.unwrap", "neptune_ir.as_tensor", "neptune_ir.from_tensor"):
                # aliases of the same buffer (:131-159; as_tensor / from_tensor stay casts, :705-733); only the logical
                # origin may change
                src: Buffer = env[op.operands[0]]
                dst_ty = op.types[1]
                if isinstance(dst_ty, TempType):
                    if tuple(src.data.shape) != dst_ty.bounds.shape:
                        raise OracleError(
                            f"{n}: buffer shape {src.data.shape} does not match {dst_ty.bounds.shape} "
                            "(memref.cast ?->static would fail)")
                    env[op.results[0]] = Buffer(src.data, dst_ty.bounds.lb)
                else:
                    env[op.results[0]] = Buffer(src.data, tuple([0] * src.data.ndim))
            elif n == "neptune_ir.apply":
                env[op.results[0]] = self._apply(op, [env[o] for o in op.operands])
            elif n in ("neptune_ir.apply_linear", "neptune_ir.apply_nonlinear"):
                callee = self.funcs.get(op.attrs["callee"])
                if callee is None:
                    raise OracleError(f"unresolved symbol @{op.attrs['callee']}")
                outs = self._run_function(callee, [env[o] for o in op.operands])
                for r, v in zip(op.results, outs):
                    env[r] = v
            elif n == "neptune_ir.store":
                self._store(op, env[op.operands[0]], env[op.operands[1]])
            elif n in ("neptune_ir.return", "func.return", "return"):
                return [env[o] for o in op.operands]
            elif n == "arith.constant":
                env[op.results[0]] = _const(op)
            elif n == "neptune_ir.reduce":
                env[op.results[0]] = self._reduce(op, env[op.operands[0]])
            elif n == "neptune_ir.time_advance" and not op.attrs.get("opaque"):
                # explicit method: k = rhs(state); out = state + dt * k  (mulf, then addf;
                # lib/Passes/HighLevelConvertion.cpp:77-120), over the whole box
                state: Buffer = env[op.operands[0]]
                dt = env[op.operands[1]]
                callee = self.funcs.get(op.attrs["rhs"][1])
                if callee is None:
                    raise OracleError("time_advance: unresolved rhs symbol")
                k = self._run_function(callee, [state])[0]
                dt_k = np.multiply(np.float64(dt), k.data)
                env[op.results[0]] = Buffer(np.add(state.data, dt_k), state.lb)
            elif (n.startswith("arith.") or n.startswith("math.")) and not op.regions:
                # scalar arithmetic on reduce results / constants at function level
                self._eval_block([op], env, None, "", partial=True)
            elif op.attrs.get("opaque"):
                raise Unsupported(
                    f"{n}: outside the stencil hot path (solver / time-stepping surface); the oracle "
                    "restates apply/access/load/store/wrap/unwrap/opdef/call only")
            else:
                raise Unsupported(f"{n} at function level")
        return []

    def _store(self, op: Op, src: Buffer, dst: Buffer) -> None:
        b = op.attrs.get("bounds")
        if b is None:
            if src.data.shape != dst.data.shape:
                raise OracleError("store: shapes differ")
            # memref.copy (:176-179).  numpy handles src is dst (no-op) correctly.
            np.copyto(dst.data, src.data)
            return
        sl_s, sl_d = [], []
        for d in range(b.rank):
            sl_s.append(slice(b.lb[d] - src.lb[d], b.ub[d] - src.lb[d]))  # each side its own origin (:201-210)
            sl_d.append(slice(b.lb[d] - dst.lb[d], b.ub[d] - dst.lb[d]))
            for s, n_ in ((sl_s[-1], src.data.shape[d]), (sl_d[-1], dst.data.shape[d])):
                if b.ub[d] > b.lb[d] and (s.start < 0 or s.stop > n_):
                    raise OutOfBounds("store bounds leave a buffer")
        dst.data[tuple(sl_d)] = src.data[tuple(sl_s)]

    # ---- reduce {kind = "sum"} (:589-698): serial left-to-right sum in row-major order -----------
    def _reduce(self, op: Op, src: Buffer):
        kind = op.attrs.get("kind")
        if kind != "sum":
            raise Unsupported('MVP reduce only supports kind="sum"')
        b = op.attrs.get("bounds")
        rank = src.data.ndim
        lb = b.lb if isinstance(b, Bounds) else src.lb
        ub = b.ub if isinstance(b, Bounds) else tuple(l + n for l, n in zip(src.lb, src.data.shape))
        sl = []
        for d in range(rank):
            lo, hi = lb[d] - src.lb[d], ub[d] - src.lb[d]
            if hi > lo and (lo < 0 or hi > src.data.shape[d]):
                raise OutOfBounds("reduce bounds leave the input buffer")
            sl.append(slice(lo, hi))
        x = np.ascontiguousarray(src.data[tuple(sl)]).reshape(-1)
        dt = src.data.dtype.type
        if x.size == 0:
            return dt(0)
        # acc = 0; acc = acc + x[0]; acc = acc + x[1]; ...  np.cumsum is exactly this sequential chain
        return np.cumsum(np.concatenate([np.zeros(1, dt), x]), dtype=dt)[-1]

    # ---- apply (:258-448) ---------------------------------------------------------------
    def _apply(self, op: Op, inputs: List[Buffer]) -> Buffer:
        bounds: Bounds = op.attrs.get("bounds")
        if not isinstance(bounds, Bounds):
            raise OracleError("apply: missing required 'bounds' attribute")
        rank = bounds.rank
        if rank == 0:
            raise OracleError("0-D apply not supported")
        res_ty: TempType = op.types[1]
        blk = op.regions[0]
        if len(blk.args) != rank + len(inputs):
            raise OracleError(
                f"apply-like region block arg count must be (bounds rank + number of inputs) = "
                f"{rank + len(inputs)}, but got {len(blk.args)}")
        for d in range(rank):
            if not (isinstance(blk.args[d][1], ScalarType) and blk.args[d][1].name == "index"):
                raise OracleError(f"region arg #{d} must be index")
        in_types = op.types[0]
        for k in range(len(inputs)):
            if blk.args[rank + k][1] != in_types[k]:
                raise OracleError(f"region input arg #{rank + k} type mismatch")
        if inputs[0].data.shape != res_ty.bounds.shape:
            raise OracleError("apply: result shape must equal input 0's shape (copy-through cast)")
        # 1. fresh result  2. copy-through of input 0, physical-index-wise
        out = np.array(inputs[0].data, dtype=np_dtype(res_ty.element), copy=True, order="C")
        result = Buffer(out, res_ty.bounds.lb)
        if any(u <= l for l, u in zip(bounds.lb, bounds.ub)):
            return result
        pts = PointSet(bounds.lb, bounds.ub)
        env: Dict[str, object] = {}
        for d in range(rank):
            env[blk.args[d][0]] = ("index", d)
        for k, b in enumerate(inputs):
            env[blk.args[rank + k][0]] = b
        val = self._eval_block(blk.ops, env, pts, res_ty.element)
        sl = []
        for d in range(rank):
            lo, hi = bounds.lb[d] - result.lb[d], bounds.ub[d] - result.lb[d]  # stored at p - out_lb (:427-444)
            if lo < 0 or hi > out.shape[d]:
                raise OutOfBounds("apply.bounds leave the result box")
            sl.append(slice(lo, hi))
        out[tuple(sl)] = np.broadcast_to(val, pts.shape)
        return result

    def _value(self, env, name: str, pts: PointSet):
        v = env[name]
        if isinstance(v, tuple) and v[0] == "index":
            return pts.index(v[1])
        return v

    def _eval_block(self, ops: List[Op], env: Dict[str, object], pts: PointSet, elem: str, partial: bool = False):
        """evaluate ops in textual order; returns the yielded value(s) (first one).
        partial=True: a run of plain ops without terminator (function-level scalar arithmetic)"""
        for op in ops:
            n = op.name
            if n == "neptune_ir.access":
                buf = env[op.operands[0]]
                if not isinstance(buf, Buffer):
                    raise OracleError("access input must be a temp")
                env[op.results[0]] = pts.access(buf, op.attrs["offsets"])
            elif n == "arith.constant":
                env[op.results[0]] = _const(op)
            elif n in _BINF:
                a, b = (self._value(env, o, pts) for o in op.operands)
                dt = np_dtype(op.types[0].name)
                with np.errstate(all="ignore"):
                    r = _BINF[n](a, b)
                if r.dtype != dt:
                    raise OracleError(f"{n}: operand dtype {r.dtype} != {dt}")
                env[op.results[0]] = r
            elif n == "arith.negf":
                env[op.results[0]] = np.negative(self._value(env, op.operands[0], pts))
            elif n in ("arith.maximumf", "arith.minimumf"):
                a, b = (self._value(env, o, pts) for o in op.operands)
                env[op.results[0]] = (np.maximum if n == "arith.maximumf" else np.minimum)(a, b)
            elif n in ("arith.maxnumf", "arith.minnumf"):
                a, b = (self._value(env, o, pts) for o in op.operands)
                env[op.results[0]] = (np.fmax if n == "arith.maxnumf" else np.fmin)(a, b)
            elif n == "math.sqrt":
                with np.errstate(all="ignore"):
                    env[op.results[0]] = np.sqrt(self._value(env, op.operands[0], pts))
            elif n == "math.absf":
                env[op.results[0]] = np.abs(self._value(env, op.operands[0], pts))
            elif n in ("math.floor", "math.ceil"):
                env[op.results[0]] = (np.floor if n == "math.floor" else np.ceil)(self._value(env, op.operands[0], pts))
            elif n == "math.copysign":
                a, b = (self._value(env, o, pts) for o in op.operands)
                env[op.results[0]] = np.copysign(a, b)
            elif n in _ELEMENTARY:
                # not exactly specified (libm in the reference's lowering): parity for bodies using these is to a few ulp
                with np.errstate(all="ignore"):
                    env[op.results[0]] = _ELEMENTARY[n](*[self._value(env, o, pts) for o in op.operands])
            elif n in _BINI:
                a, b = (self._value(env, o, pts) for o in op.operands)
                env[op.results[0]] = _BINI[n](a, b)
            elif n == "arith.cmpi":
                a, b = (self._value(env, o, pts) for o in op.operands)
                pred = op.attrs["predicate"]
                if pred not in _CMPI:
                    raise Unsupported(f"cmpi predicate {pred}")
                env[op.results[0]] = _CMPI[pred](a, b)
            elif n == "arith.cmpf":
                a, b = (self._value(env, o, pts) for o in op.operands)
                pred = op.attrs["predicate"]
                if pred not in _CMPF:
                    raise Unsupported(f"cmpf predicate {pred}")
                env[op.results[0]] = _CMPF[pred](a, b)
            elif n == "arith.select":
                c, a, b = (self._value(env, o, pts) for o in op.operands)
                env[op.results[0]] = np.where(c, a, b)
            elif n == "arith.index_cast":
                dst = op.types[1] if len(op.types) > 1 else op.types[0]
                env[op.results[0]] = np.asarray(self._value(env, op.operands[0], pts)).astype(np_dtype(dst.name))
            elif n in ("arith.sitofp", "arith.uitofp"):
                dst = op.types[1]
                env[op.results[0]] = np.asarray(self._value(env, op.operands[0], pts)).astype(np_dtype(dst.name))
            elif n in ("arith.extf", "arith.truncf"):
                dst = op.types[1]
                env[op.results[0]] = np.asarray(self._value(env, op.operands[0], pts)).astype(np_dtype(dst.name))
            elif n == "scf.if":
                env_res = self._eval_if(op, env, pts, elem)
                for r, v in zip(op.results, env_res):
                    env[r] = v
            elif n in ("neptune_ir.yield", "scf.yield"):
                vals = [self._value(env, o, pts) for o in op.operands]
                if n == "neptune_ir.yield":
                    if len(vals) != 1:
                        raise OracleError("MVP: only single-scalar yield is supported")
                    if np.asarray(vals[0]).dtype != np_dtype(elem):
                        raise OracleError("yield operand type must equal apply result element type")
                    return vals[0]
                return vals
            else:
                raise Unsupported(f"{n} inside an apply region")
        if partial:
            return None
        raise OracleError("region has no terminator")

    def _eval_if(self, op: Op, env, pts: PointSet, elem: str):
        cond = np.broadcast_to(np.asarray(self._value(env, op.operands[0], pts)), pts.shape)
        if len(op.regions) < 2 and op.results:
            raise OracleError("scf.if with results needs an else region")
        g = pts.to_gather()
        flat = cond.reshape(-1)
        n_out = len(op.results)
        outs: List[Optional[np.ndarray]] = [None] * n_out
        for branch, mask in ((0, flat), (1, ~flat)):
            if branch >= len(op.regions) or not mask.any():
                continue
            sub = g.subset(mask)
            # values defined outside the if must be narrowed to the sub point set
            sub_env = _NarrowEnv(env, pts, mask)
            vals = self._eval_block(op.regions[branch].ops, sub_env, sub, elem)
            for r in range(n_out):
                v = np.broadcast_to(np.asarray(vals[r]), (int(mask.sum()),))
                if outs[r] is None:
                    outs[r] = np.empty(flat.shape, dtype=v.dtype)
                outs[r][mask] = v
        return [o.reshape(pts.shape) if o is not None else None for o in outs]


class _NarrowEnv(dict):
    """view of an outer environment restricted to the points selected by `mask`"""

    def __init__(self, outer, pts: PointSet, mask: np.ndarray):
        super().__init__()
        self._outer = outer
        self._pts = pts
        self._mask = mask

    def __missing__(self, key):
        v = self._outer[key]
        if isinstance(v, (Buffer, tuple)) or np.ndim(v) == 0:
            nv = v
        else:
            nv = np.broadcast_to(v, self._pts.shape).reshape(-1)[self._mask]
        self[key] = nv
        return nv


# --------------------------------------------------------------------------------------
# convenience: faithful scalar loop (small cases only) used to cross-check the vectorised path
# --------------------------------------------------------------------------------------
def apply_scalar_loops(module: Module, sym: str, *arrays: np.ndarray) -> np.ndarray:
    """Evaluate opdef `@sym` with the reference's literal loop structure: one point at a time,
    row-major order (DataflowLowering.cpp:289-308).  Pure-Python speed: tiny inputs only."""
    f = module.funcs[sym]
    args = [module._bind_arg(a, ty) for a, ty in zip(arrays, f.arg_types)]
    env: Dict[str, object] = {name: val for (name, _), val in zip(f.body.args, args)}
    for op in f.body.ops:
        if op.name == "neptune_ir.apply":
            inputs = [env[o] for o in op.operands]
            bounds: Bounds = op.attrs["bounds"]
            res_ty: TempType = op.types[1]
            out = np.array(inputs[0].data, copy=True)
            blk = op.regions[0]
            rank = bounds.rank
            import itertools
            for p in itertools.product(*[range(l, u) for l, u in zip(bounds.lb, bounds.ub)]):
                pts = PointSet(p, tuple(x + 1 for x in p))
                benv: Dict[str, object] = {}
                for d in range(rank):
                    benv[blk.args[d][0]] = ("index", d)
                for k, b in enumerate(inputs):
                    benv[blk.args[rank + k][0]] = b
                v = module._eval_block(blk.ops, benv, pts, res_ty.element)
                out[tuple(x - l for x, l in zip(p, res_ty.bounds.lb))] = np.asarray(v).reshape(-1)[0]
            env[op.results[0]] = Buffer(out, res_ty.bounds.lb)
        elif op.name in ("neptune_ir.apply_linear", "neptune_ir.apply_nonlinear"):
            outs = apply_scalar_loops(module, op.attrs["callee"], *[env[o].data for o in op.operands])
            env[op.results[0]] = Buffer(outs, op.types[1][0].bounds.lb)
        elif op.name == "neptune_ir.return":
            return env[op.operands[0]].data
        else:
            raise Unsupported(op.name)
    raise OracleError("no return")

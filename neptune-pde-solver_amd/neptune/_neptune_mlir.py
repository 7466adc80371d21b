"""IR builder behind the Python DSL: `Compiler` / `Value` with the method surface of the
reference's pybind11 module `_neptune_mlir` (python_frontend/bindings/NeptuneModule.cpp:8-35, backed
by lib/Compiler/NeptuneCompiler.cpp), re-implemented without MLIR: it assembles NeptuneIR *text*
in the reference's ODS assembly formats and hands it to the HIP lowering
(libneptune_lowering.so) instead of the LLVM pipeline.

Differences from the reference builder, all deliberate:
  * create_apply emits the region signature the reference's own verifier requires -- `rank` index
    arguments followed by one temp per input (NeptuneIRVerifier.cpp:150-168).  The reference
    builder adds only the temp arguments (NeptuneCompiler.cpp:136-140), so its output fails
    ApplyOp::verify; body callbacks still receive just the input temps, exactly as there.
  * compile_to_object_file produces a gfx950 HIP object (hipcc -c) rather than an x86 one.
"""
from __future__ import annotations

import os
import subprocess
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence


@dataclass(frozen=True)
class _Type:
    kind: str                      # "temp" | "field" | "memref" | "scalar" | "none"
    elem: str = "f64"
    lb: tuple = ()
    ub: tuple = ()
    location: str = "cell"
    rank: int = 0

    def text(self) -> str:
        if self.kind in ("temp", "field"):
            lb = ", ".join(str(x) for x in self.lb)
            ub = ", ".join(str(x) for x in self.ub)
            return (f"!neptune_ir.{self.kind}<element = {self.elem}, bounds = #neptune_ir.bounds<lb = [{lb}], "
                    f"ub = [{ub}]>, location = #neptune_ir.location<\"{self.location}\">>")
        if self.kind == "memref":
            return "memref<" + "x".join(["?"] * self.rank + [self.elem]) + ">"
        if self.kind == "scalar":
            return self.elem
        return "none"


class Value:
    """opaque SSA value handle (reference: PyValue, include/Frontend/NeptuneCompiler.h)"""

    def __init__(self, name: Optional[str], type_: _Type):
        self.name = name
        self.type = type_

    def __repr__(self) -> str:
        return f"<Value {self.name} : {self.type.text()}>" if self.name else "<Value null>"

    __str__ = __repr__


class _Scope:
    def __init__(self, indent: int):
        self.lines: List[str] = []
        self.indent = indent


class Compiler:
    def __init__(self):
        self._top: List[str] = []          # module-level ops, in creation order
        self._scopes: List[_Scope] = []
        self._counter = 0
        self._func: Optional[dict] = None

    # ---- helpers -----------------------------------------------------------------------
    def _fresh(self, hint: str = "") -> str:
        self._counter += 1
        return f"%{hint}{self._counter}"

    def _emit(self, line: str) -> None:
        if not self._scopes:
            # module level, like the reference builder whose insertion point starts in the module body
            # (the reference's test_user.py calls assemble_matrix there); such ops are not lowerable
            self._top.append("  " + line)
            return
        sc = self._scopes[-1]
        sc.lines.append(" " * sc.indent + line)

    @staticmethod
    def _temp(lb: Sequence[int], ub: Sequence[int], loc: str = "cell", elem: str = "f64", kind: str = "temp") -> _Type:
        return _Type(kind, elem, tuple(int(x) for x in lb), tuple(int(x) for x in ub), loc, len(lb))

    # ---- basic ops (NeptuneModule.cpp:13-20) --------------------------------------------
    def dump(self) -> str:
        body = "\n".join(self._top)
        return "module {\n" + body + ("\n" if body else "") + "}\n"

    def create_wrap(self, buffer: Optional[Value], type_hint) -> Value:
        """buffer: a memref Value; type_hint: (lb, ub[, location]) of the field, or anything else for
        the reference's placeholder behaviour (it returns the buffer unchanged, NeptuneCompiler.cpp:57-71)"""
        if isinstance(buffer, Value) and buffer.type.kind == "memref" and isinstance(type_hint, (tuple, list)):
            lb, ub = type_hint[0], type_hint[1]
            loc = type_hint[2] if len(type_hint) > 2 else "cell"
            ft = self._temp(lb, ub, loc, buffer.type.elem, "field")
            name = self._fresh("f")
            self._emit(f"{name} = neptune_ir.wrap {buffer.name} : {buffer.type.text()} -> {ft.text()}")
            return Value(name, ft)
        return buffer if isinstance(buffer, Value) else Value(None, _Type("none"))

    def create_load(self, field: Value) -> Value:  # extension: the reference builder has no load/store/unwrap
        tt = _Type("temp", field.type.elem, field.type.lb, field.type.ub, field.type.location, field.type.rank)
        name = self._fresh("t")
        self._emit(f"{name} = neptune_ir.load {field.name} : {field.type.text()} -> {tt.text()}")
        return Value(name, tt)

    def create_store(self, value: Value, field: Value, lb=None, ub=None) -> None:
        attr = ""
        if lb is not None:
            attr = " {bounds = #neptune_ir.bounds<lb = [%s], ub = [%s]>}" % (", ".join(map(str, lb)), ", ".join(map(str, ub)))
        self._emit(f"neptune_ir.store {value.name} to {field.name}{attr} : {value.type.text()} to {field.type.text()}")

    def create_unwrap(self, field: Value) -> Value:
        mt = _Type("memref", field.type.elem, rank=field.type.rank)
        name = self._fresh("m")
        self._emit(f"{name} = neptune_ir.unwrap {field.name} : {field.type.text()} -> {mt.text()}")
        return Value(name, mt)

    def create_access(self, temp: Value, offsets: Sequence[int]) -> Value:
        if temp.type.kind != "temp":
            raise TypeError("create_access needs a temp value")
        if len(offsets) != temp.type.rank:
            raise ValueError(f"access needs {temp.type.rank} offsets, got {len(offsets)}")
        name = self._fresh("a")
        offs = ", ".join(str(int(o)) for o in offsets)
        self._emit(f"{name} = neptune_ir.access {temp.name}[{offs}] : {temp.type.text()} -> {temp.type.elem}")
        return Value(name, _Type("scalar", temp.type.elem))

    def _arith(self, op: str, lhs: Value, rhs: Value) -> Value:
        if lhs.type != rhs.type or lhs.type.kind != "scalar":
            raise TypeError(f"{op}: operands must be scalars of one type, got {lhs.type.text()} and {rhs.type.text()}")
        name = self._fresh("v")
        self._emit(f"{name} = arith.{op} {lhs.name}, {rhs.name} : {lhs.type.elem}")
        return Value(name, lhs.type)

    def create_arith_add(self, lhs: Value, rhs: Value) -> Value:
        return self._arith("addf", lhs, rhs)

    def create_arith_sub(self, lhs: Value, rhs: Value) -> Value:
        return self._arith("subf", lhs, rhs)

    def create_arith_mul(self, lhs: Value, rhs: Value) -> Value:
        return self._arith("mulf", lhs, rhs)

    def create_arith_div(self, lhs: Value, rhs: Value) -> Value:  # extension (the reference DSL has no '/')
        return self._arith("divf", lhs, rhs)

    def create_constant(self, value: float) -> Value:
        name = self._fresh("c")
        self._emit(f"{name} = arith.constant {float(value)!r} : f64")
        return Value(name, _Type("scalar", "f64"))

    # ---- DSL core (NeptuneModule.cpp:21-27) -----------------------------------------------
    def create_apply(self, inputs: Sequence[Value], lb: Sequence[int], ub: Sequence[int],
                     body_builder: Callable[[List[Value]], Value]) -> Value:
        if not inputs:
            raise ValueError("apply needs at least one input (it is the copy-through source)")
        for v in inputs:
            if v.type.kind != "temp":
                raise TypeError("apply inputs must be temps")
        rank = len(lb)
        if len(ub) != rank or any(v.type.rank != rank for v in inputs):
            raise ValueError("apply bounds / input rank mismatch")
        res_t = inputs[0].type                      # result type = first input's type (NeptuneCompiler.cpp:125-127)
        res = self._fresh("r")
        in_names = ", ".join(v.name for v in inputs)
        in_types = ", ".join(v.type.text() for v in inputs)
        b = "#neptune_ir.bounds<lb = [%s], ub = [%s]>" % (", ".join(str(int(x)) for x in lb), ", ".join(str(int(x)) for x in ub))
        self._emit(f"{res} = neptune_ir.apply({in_names}) attributes {{bounds = {b}}}")
        self._emit(f"  : ({in_types}) -> {res_t.text()} {{")
        idx = [self._fresh("i") for _ in range(rank)]
        args = [Value(self._fresh("in"), v.type) for v in inputs]
        sig = ", ".join(f"{i}: index" for i in idx) + ", " + ", ".join(f"{a.name}: {a.type.text()}" for a in args)
        outer = self._scopes[-1]
        self._emit(f"^bb0({sig}):")
        self._scopes.append(_Scope(outer.indent + 2))
        out = body_builder(list(args))
        if not isinstance(out, Value) or out.type != _Type("scalar", res_t.elem):
            raise TypeError("apply body must return a scalar Value of the result's element type")
        self._emit(f"neptune_ir.yield {out.name} : {res_t.elem}")
        inner = self._scopes.pop()
        outer.lines.extend(inner.lines)
        self._emit("}")
        return Value(res, res_t)

    def create_linear_opdef(self, name: str, lb: Sequence[int], ub: Sequence[int], loc_kind: str,
                            body_builder: Callable[[List[Value]], Value]) -> None:
        """(temp) -> temp linear operator over an f64 temp of box [lb,ub) (NeptuneCompiler.cpp:160-203)"""
        self._opdef("linear_opdef", name, lb, ub, loc_kind, body_builder)

    def create_nonlinear_opdef(self, name, lb, ub, loc_kind, body_builder, num_inputs: int = 1) -> None:  # extension
        self._opdef("nonlinear_opdef", name, lb, ub, loc_kind, body_builder, num_inputs)

    def _opdef(self, kind, name, lb, ub, loc_kind, body_builder, num_inputs: int = 1) -> None:
        if self._scopes:
            raise RuntimeError("opdefs are module-level: finish the current function first")
        t = self._temp(lb, ub, loc_kind)
        args = [Value(self._fresh("arg"), t) for _ in range(num_inputs)]
        self._scopes.append(_Scope(4))
        out = body_builder(list(args))
        if not isinstance(out, Value) or out.type != t:
            raise TypeError(f"{kind} body must return a temp of the operator's type")
        self._emit(f"neptune_ir.return {out.name} : {t.text()}")
        sc = self._scopes.pop()
        tys = ", ".join([t.text()] * num_inputs)
        self._top.append(f"  neptune_ir.{kind} @{name} : ({tys}) -> {t.text()} {{")
        self._top.append("  ^bb0(" + ", ".join(f"{a.name}: {t.text()}" for a in args) + "):")
        self._top.extend(sc.lines)
        self._top.append("  }")

    def create_apply_linear(self, symbol: str, inputs: Sequence[Value], result_like: Optional[Value] = None) -> Value:  # extension
        res_t = (result_like or inputs[0]).type
        name = self._fresh("y")
        tys = ", ".join(v.type.text() for v in inputs)
        self._emit(f"{name} = neptune_ir.apply_linear @{symbol}({', '.join(v.name for v in inputs)}) : ({tys}) -> {res_t.text()}")
        return Value(name, res_t)

    # ---- extensions on the lowered path: reduce and explicit time stepping -----------------
    def create_reduce_sum(self, temp: Value, lb=None, ub=None) -> Value:
        """neptune_ir.reduce %t [in bounds] {kind = "sum"} (NeptuneIROps.td:272-299): a scalar of the element type.
        Of a single-use apply result it lowers to ONE kernel (dot products, norms)."""
        if temp.type.kind != "temp":
            raise TypeError("create_reduce_sum needs a temp value")
        name = self._fresh("s")
        where = ""
        if lb is not None:
            where = " in #neptune_ir.bounds<lb = [%s], ub = [%s]>" % (", ".join(map(str, lb)), ", ".join(map(str, ub)))
        self._emit(f'{name} = neptune_ir.reduce {temp.name}{where} {{kind = "sum"}} : {temp.type.text()} -> {temp.type.elem}')
        return Value(name, _Type("scalar", temp.type.elem))

    def create_time_advance_explicit(self, state: Value, dt: float, rhs_symbol: str) -> Value:
        """neptune_ir.time_advance %state, %dt {method = 0, rhs = @symbol}: state + dt * rhs(state), one fused kernel
        when @symbol is a single apply of the state (NeptuneIROps.td:745-775; explicit form of HighLevelConvertion.cpp:77-120)"""
        if state.type.kind != "temp" or state.type.elem != "f64":
            raise TypeError("explicit time_advance needs an f64 temp state (dt is f64 in the op definition)")
        dtv = self._fresh("dt")
        self._emit(f"{dtv} = arith.constant {float(dt)!r} : f64")
        name = self._fresh("u")
        self._emit(f"{name} = neptune_ir.time_advance {state.name}, {dtv} {{method = 0 : i32, rhs = @{rhs_symbol}}} "
                   f": {state.type.text()}, f64 -> {state.type.text()}")
        return Value(name, state.type)

    # ---- solver surface: emitted textually, never lowered here (host PETSc path) -----------
    def create_assemble_matrix(self, op_symbol: str) -> Value:
        name = self._fresh("A")
        self._emit(f"{name} = neptune_ir.assemble_matrix @{op_symbol} : memref<?x?xf64>")
        return Value(name, _Type("memref", "f64", rank=2))

    def create_solve_linear(self, matrix: Value, rhs: Value, solver: str = "cg", tol: float = 1e-6) -> Value:
        name = self._fresh("x")
        self._emit(f"{name} = neptune_ir.solve_linear {matrix.name}, {rhs.name} {{solver = \"{solver}\", tol = {float(tol)!r} : f64}} "
                   f": {matrix.type.text()}, {rhs.type.text()} -> {rhs.type.text()}")
        return Value(name, rhs.type)

    # ---- functions (NeptuneModule.cpp:30-33) ----------------------------------------------
    def start_function(self, name: str, arg_type_hints: Sequence) -> None:
        """arg_type_hints: Values whose types the arguments take (reference behaviour), or
        ("memref", rank[, elem]) / ("temp", lb, ub[, loc]) tuples."""
        if self._func is not None or self._scopes:
            raise RuntimeError("already inside a function")
        types: List[_Type] = []
        for h in arg_type_hints:
            if isinstance(h, Value):
                types.append(h.type if h.type.kind != "none" else _Type("scalar", "f64"))
            elif isinstance(h, (tuple, list)) and h and h[0] == "memref":
                types.append(_Type("memref", h[2] if len(h) > 2 else "f64", rank=int(h[1])))
            elif isinstance(h, (tuple, list)) and h and h[0] in ("temp", "field"):
                types.append(self._temp(h[1], h[2], h[3] if len(h) > 3 else "cell", kind=h[0]))
            else:
                types.append(_Type("scalar", "f64"))  # reference fallback (NeptuneCompiler.cpp:247-250)
        args = [Value(f"%arg{i}", t) for i, t in enumerate(types)]
        self._func = {"name": name, "args": args, "ret": None}
        self._scopes.append(_Scope(4))

    def get_function_arg(self, index: int) -> Value:
        if self._func is None:
            raise RuntimeError("Not inside a function!")
        return self._func["args"][index]

    def create_return(self, value: Value) -> None:
        if self._func is None:
            raise RuntimeError("Not inside a function!")
        self._func["ret"] = value
        self._emit(f"func.return {value.name} : {value.type.text()}")

    def end_function(self) -> None:
        if self._func is None:
            raise RuntimeError("Not inside a function!")
        f, sc = self._func, self._scopes.pop()
        sig = ", ".join(f"{a.name}: {a.type.text()}" for a in f["args"])
        ret = f" -> {f['ret'].type.text()}" if f["ret"] is not None else ""
        if f["ret"] is None:
            sc.lines.append("    func.return")
        self._top.append(f"  func.func @{f['name']}({sig}){ret} {{")
        self._top.extend(sc.lines)
        self._top.append("  }")
        self._func = None

    # ---- AOT --------------------------------------------------------------------------------
    def compile_to_object_file(self, path: str) -> None:
        """lower the module and compile it to a relocatable gfx950 HIP object at `path`
        (reference: NeptuneCompiler::compileToObjectFile, NeptuneCompiler.cpp:304-358 -- pipeline,
        LLVM IR, TargetMachine; here: HIP lowering + hipcc -c)."""
        from neptune_hip import _capi, lowering
        src, _ = lowering.to_hip(self.dump())
        hip_path = str(path) + ".hip"
        with open(hip_path, "w") as fh:
            fh.write(src)
        cc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [cc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-x", "hip", "-c", hip_path,
               "-I", str(_capi.REPO_ROOT), "-o", str(path)]
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError("Failed to compile the lowered module:\n" + p.stderr)

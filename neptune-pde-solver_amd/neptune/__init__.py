"""neptune -- Python frontend of the MI355X backend, API-compatible with the reference's
python_frontend/neptune package (python_frontend/neptune/__init__.py:12-44): same names, same
decorators, same tracing model; the IR it builds is lowered to HIP instead of LLVM."""
from .core import GlobalContext as Context
from .core import get_compiler, reset
from .expr import Expr
from .dsl import apply, stencil, linear_op_def, assemble_matrix, solve_linear
from .backend import jit_compile
from .jit import jit_class


# ---- helpers for function bodies (extensions; the reference has no Python spelling for these) ----
def wrap(buffer: Expr, bounds, location="cell") -> Expr:
    """neptune_ir.wrap: view a memref argument as a field over the logical box bounds=(lb, ub)"""
    return Expr(get_compiler().create_wrap(buffer._handle, (bounds[0], bounds[1], location)))


def load(field: Expr) -> Expr:
    return Expr(get_compiler().create_load(field._handle))


def store(value: Expr, field: Expr, bounds=None) -> None:
    lb, ub = bounds if bounds is not None else (None, None)
    get_compiler().create_store(value._handle, field._handle, lb, ub)


def unwrap(field: Expr) -> Expr:
    return Expr(get_compiler().create_unwrap(field._handle))


def apply_linear(symbol: str, *inputs: Expr) -> Expr:
    return Expr(get_compiler().create_apply_linear(symbol, [i._handle for i in inputs]))


def reduce_sum(value: Expr, bounds=None) -> Expr:
    """sum of a temp over its box (or the sub-box bounds=(lb, ub)); `reduce_sum(apply(...)(kernel))` is one kernel"""
    lb, ub = bounds if bounds is not None else (None, None)
    return Expr(get_compiler().create_reduce_sum(value._handle, lb, ub))


def time_advance(state: Expr, dt: float, rhs: str) -> Expr:
    """explicit Euler step state + dt * rhs(state); `rhs` is the symbol a linear_op_def returned"""
    return Expr(get_compiler().create_time_advance_explicit(state._handle, dt, rhs))


__all__ = ["Context", "get_compiler", "reset", "Expr", "apply", "stencil", "linear_op_def", "assemble_matrix",
           "solve_linear", "jit_compile", "jit_class", "wrap", "load", "store", "unwrap", "apply_linear", "reduce_sum", "time_advance"]

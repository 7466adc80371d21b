"""Tracing values of the DSL.  An `Expr` wraps one SSA value of the module under construction; indexing it
with stencil offsets emits `neptune_ir.access`, arithmetic between Exprs (or an Expr and a Python number,
which becomes an `arith.constant`) emits the matching `arith` op through the active builder.

Behavioural counterpart of the reference tracer (python_frontend/neptune/expr.py:20-58: `__getitem__` ->
access, `+ - *` -> arith); `/` is an extension of this package (the reference has no division)."""
import numbers

from .core import get_compiler

# Python operator name -> builder method emitting the op
_ARITH = {"add": "create_arith_add", "sub": "create_arith_sub", "mul": "create_arith_mul", "truediv": "create_arith_div"}


class Expr:
    __slots__ = ("_handle",)

    def __init__(self, handle):
        self._handle = handle

    @staticmethod
    def lift(value) -> "Expr":
        """an Expr as is; a real number as a constant of the element type"""
        if isinstance(value, Expr):
            return value
        if isinstance(value, numbers.Real) and not isinstance(value, bool):
            return Expr(get_compiler().create_constant(float(value)))
        raise TypeError(f"cannot use {type(value).__name__} in a stencil expression (expected Expr or a real number)")

    def __getitem__(self, offsets):
        """u[-1], u[0, 1], ...: the value of this temp at a constant offset from the current point"""
        if isinstance(offsets, numbers.Integral):
            offsets = (offsets,)
        if not isinstance(offsets, (tuple, list)) or not all(isinstance(o, numbers.Integral) for o in offsets):
            raise TypeError(f"stencil offsets must be integers, got {offsets!r}")
        return Expr(get_compiler().create_access(self._handle, [int(o) for o in offsets]))

    def __repr__(self):
        return f"Expr({self._handle!r})"


def _install_operator(name: str, builder_method: str) -> None:
    def forward(self, other):
        rhs = Expr.lift(other)
        return Expr(getattr(get_compiler(), builder_method)(self._handle, rhs._handle))

    def reflected(self, other):
        lhs = Expr.lift(other)
        return Expr(getattr(get_compiler(), builder_method)(lhs._handle, self._handle))

    forward.__name__, reflected.__name__ = f"__{name}__", f"__r{name}__"
    setattr(Expr, forward.__name__, forward)
    setattr(Expr, reflected.__name__, reflected)


for _name, _method in _ARITH.items():
    _install_operator(_name, _method)

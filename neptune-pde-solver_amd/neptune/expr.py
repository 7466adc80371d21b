"""Operator-overloading tracer (mirror of python_frontend/neptune/expr.py:3-58): indexing a temp
emits neptune_ir.access, + - * emit arith ops on the builder; numbers become arith.constant."""
from .core import get_compiler


class Expr:
    def __init__(self, handle):
        self._handle = handle

    def _get_compiler(self):
        return get_compiler()

    def _as_expr(self, other):
        if isinstance(other, Expr):
            return other
        if isinstance(other, (int, float)):
            return Expr(self._get_compiler().create_constant(float(other)))
        raise TypeError(f"Unsupported operand type: {type(other)}")

    # u[-1], u[0, 1], ...
    def __getitem__(self, index):
        if isinstance(index, int):
            offsets = [index]
        elif isinstance(index, (tuple, list)):
            offsets = list(index)
        else:
            raise TypeError(f"Indices must be integers or tuples, got {type(index)}")
        return Expr(self._get_compiler().create_access(self._handle, offsets))

    def __add__(self, other):
        other = self._as_expr(other)
        return Expr(self._get_compiler().create_arith_add(self._handle, other._handle))

    def __sub__(self, other):
        other = self._as_expr(other)
        return Expr(self._get_compiler().create_arith_sub(self._handle, other._handle))

    def __mul__(self, other):
        other = self._as_expr(other)
        return Expr(self._get_compiler().create_arith_mul(self._handle, other._handle))

    def __truediv__(self, other):  # extension: the reference tracer has no division (expr.py:33-58)
        other = self._as_expr(other)
        return Expr(self._get_compiler().create_arith_div(self._handle, other._handle))

    def __radd__(self, other):
        return self._as_expr(other) + self

    def __rsub__(self, other):
        return self._as_expr(other) - self

    def __rmul__(self, other):
        return self._as_expr(other) * self

    def __rtruediv__(self, other):
        return self._as_expr(other) / self

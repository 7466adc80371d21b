"""The process-wide builder the DSL decorators and `Expr` operators write into.

Same role as the reference's python_frontend/neptune/core.py:1-26 (a global context exposing `get_compiler()`),
but the builder behind it is this package's text-emitting `_neptune_mlir.Compiler`: the reference checkout
imports its compiled extension here and falls back to `None` when it is missing, this one always has a builder."""
from . import _neptune_mlir


class GlobalContext:
    """holds the module being built; `reset()` starts a new one (the reference keeps one per process)"""

    def __init__(self):
        self.compiler = _neptune_mlir.Compiler()

    def reset(self):
        self.compiler = _neptune_mlir.Compiler()

    def dump(self):
        return self.compiler.dump()


_context = GlobalContext()


def get_compiler():
    return _context.compiler


def reset():
    _context.reset()

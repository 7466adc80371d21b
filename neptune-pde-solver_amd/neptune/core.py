"""Global compiler context (mirror of the reference's python_frontend/neptune/core.py:1-26).
The builder behind it is the text-emitting `_neptune_mlir.Compiler` of this package, so unlike the
reference checkout (whose compiled extension is absent) `get_compiler()` is never None here."""
from . import _neptune_mlir as _backend


class GlobalContext:
    def __init__(self):
        self.compiler = _backend.Compiler()

    def dump(self):
        return self.compiler.dump()

    def reset(self):
        """start a fresh module (extension: the reference keeps one module per process)"""
        self.compiler = _backend.Compiler()


_default_ctx = GlobalContext()


def get_compiler():
    return _default_ctx.compiler


def reset():
    _default_ctx.reset()

"""@jit_class: trace a Python class once, compile its methods, call them with arrays
(mirror of python_frontend/neptune/jit.py:8-155).

The reference wrapper traces __init__ (collecting `self.<name> = Expr` state such as an assembled
matrix) and then one method, and passes raw arguments to ctypes (NumPy marshaling is a TODO there,
jit.py:142-144).  On the stencil hot path there is no solver state, so this wrapper traces the
requested method with one argument per array, declares `memref` arguments of the arrays' ranks,
and lets LoweredModule.call marshal NumPy arrays / CUDA tensors.  A method body works with fields:

    @neptune.jit_class
    class Heat:
        def __init__(self, n): self.n = n
        def step(self, out, u):                       # memref arguments
            fo, fu = wrap(out, ...), wrap(u, ...)     # neptune.field(...) helpers, see __init__.py
            ...
"""
import functools

import numpy as np

from .backend import jit_compile
from .core import get_compiler, reset
from .expr import Expr


class JITClassWrapper:
    def __init__(self, cls, *args, **kwargs):
        self._cls = cls
        self._instance = cls(*args, **kwargs)
        self._modules = {}

    def _compile(self, method_name, sample_args):
        print(f"[Neptune JIT] Tracing {self._cls.__name__}.{method_name}...")
        reset()
        compiler = get_compiler()
        pre = getattr(self._instance, "define_operators", None)
        if callable(pre):
            pre()                                   # opdefs are module-level: define them first
        hints = []
        for a in sample_args:
            t = getattr(a, "tensor", a)
            rank = t.ndim if isinstance(t, np.ndarray) else t.dim()
            elem = "f32" if str(t.dtype).endswith("float32") else "f64"
            hints.append(("memref", rank, elem))
        fname = f"{self._cls.__name__}_{method_name}"
        compiler.start_function(fname, hints)
        args = [Expr(compiler.get_function_arg(i)) for i in range(len(sample_args))]
        result = getattr(self._instance, method_name)(*args)
        if isinstance(result, Expr):
            compiler.create_return(result._handle)
        compiler.end_function()
        mod = jit_compile(compiler)
        print(f"[Neptune JIT] Library ready: {mod.path}")
        return fname, mod

    def __getattr__(self, name):
        def method_proxy(*args):
            if name not in self._modules:
                self._modules[name] = self._compile(name, args)
            fname, mod = self._modules[name]
            print(f"[Neptune Runtime] Running {fname}...")
            return mod.call(fname, *args)

        return method_proxy


def jit_class(cls):
    @functools.wraps(cls)
    def wrapper(*args, **kwargs):
        return JITClassWrapper(cls, *args, **kwargs)

    return wrapper

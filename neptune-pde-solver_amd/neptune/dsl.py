"""DSL decorators and top-level instructions (mirror of python_frontend/neptune/dsl.py:5-74)."""
from .core import get_compiler
from .expr import Expr


def apply(inputs, bounds):
    """@neptune.apply(inputs=[u, v], bounds=([1], [9]))
    def kernel(u, v): ...        -> Expr wrapping the apply's result temp"""
    lb, ub = bounds
    compiler = get_compiler()

    def decorator(func):
        def body(arg_handles):
            result = func(*[Expr(h) for h in arg_handles])
            if not isinstance(result, Expr):
                raise TypeError(f"Kernel must return a Neptune Expr, got {type(result)}")
            return result._handle

        return Expr(compiler.create_apply([i._handle for i in inputs], lb, ub, body))

    return decorator


stencil = apply


def linear_op_def(bounds, location, name=None, apply_bounds=None):
    """Define a linear operator symbol; the scalar kernel is wrapped in one neptune_ir.apply.

    `apply_bounds` (extension) restricts the apply to a sub-box, typically the interior, so that
    neighbour accesses stay inside the field; the reference applies the kernel over the whole box
    (dsl.py:41-45), which reads out of bounds for any stencil with a non-zero offset."""
    compiler = get_compiler()

    def decorator(func):
        symbol_name = name if name else func.__name__
        lb, ub = apply_bounds if apply_bounds is not None else bounds

        def op_def_body(op_args):
            def apply_body(apply_args):
                return func(*[Expr(h) for h in apply_args])._handle

            return compiler.create_apply(op_args, lb, ub, apply_body)

        compiler.create_linear_opdef(symbol_name, bounds[0], bounds[1], location, op_def_body)
        return symbol_name

    return decorator


def assemble_matrix(op_symbol_name):
    """H = neptune.assemble_matrix("laplacian")   (solver surface: host PETSc path)"""
    return Expr(get_compiler().create_assemble_matrix(op_symbol_name))


def solve_linear(matrix, rhs, solver="cg", tol=1e-6):
    return Expr(get_compiler().create_solve_linear(matrix._handle, rhs._handle, solver, tol))

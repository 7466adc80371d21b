#!/usr/bin/env python3
"""Emit the NeptuneIR text of the stencil fixtures for any field size.

The committed fixtures under tests/mlir_tests/conversion_tests/ are this script's output at
their nominal sizes (apply-2d-5pt: 1024^2 f64, apply-3d-7pt: 512^3 f64, apply-3d-27pt: 512^3
f32).  The reference README names such a file (README.md:60-64) but ships none; the structure
follows the reference's own inputs (test/smoke_tests/smoke_time_advance.mlir:3-31,53-59,82-84):
attribute/type aliases, a linear_opdef holding one apply over the interior with region
signature (index x rank, temp x inputs), body ops in a fixed textual order, and an @entry that
does wrap, wrap, load, apply_linear, store, unwrap, return.

usage: make_stencil_mlir.py {2d5|3d7|3d27} N0 [N1 [N2]] [--time-step DT] > out.mlir
(--time-step adds @step: neptune_ir.time_advance {method = 0, rhs = @operator}; f64 kinds only)
"""
import sys

KINDS = {
    # kind: (rank, element, opdef symbol, file name, nominal size, title)
    "2d5": (2, "f64", "lap2d", "apply-2d-5pt.mlir", 1024,
            "2-D 5-point Laplacian, {dims} f64: lap = dxinv2 * ((n + s + w + e) - 4 c)"),
    "3d7": (3, "f64", "lap3d", "apply-3d-7pt.mlir", 512,
            "3-D 7-point Laplacian, {dims} f64: lap = dxinv2 * ((xm + xp + ym + yp + zm + zp) - 6 c)"),
    "3d27": (3, "f32", "lap27", "apply-3d-27pt.mlir", 512,
             "3-D 27-point stencil, {dims} f32: lap = dxinv2 * (sum of the 26 neighbours "
             "(dim-0-major order) - 26 c)"),
}


def _body(kind):
    if kind == "2d5":
        acc = [("c", (0, 0)), ("n", (-1, 0)), ("s", (1, 0)), ("w", (0, -1)), ("e", (0, 1))]
        ops = ["%four   = arith.constant 4.0 : f64",
               "%dxinv2 = arith.constant 0.125 : f64",
               "%t0  = arith.addf %n, %s : f64",
               "%t1  = arith.addf %t0, %w : f64",
               "%t2  = arith.addf %t1, %e : f64",
               "%t3  = arith.mulf %four, %c : f64",
               "%t4  = arith.subf %t2, %t3 : f64",
               "%lap = arith.mulf %dxinv2, %t4 : f64",
               "neptune_ir.yield %lap : f64"]
        return acc, ops
    if kind == "3d7":
        acc = [("c", (0, 0, 0)), ("xm", (-1, 0, 0)), ("xp", (1, 0, 0)), ("ym", (0, -1, 0)), ("yp", (0, 1, 0)),
               ("zm", (0, 0, -1)), ("zp", (0, 0, 1))]
        ops = ["%six    = arith.constant 6.0 : f64",
               "%dxinv2 = arith.constant 0.0625 : f64",
               "%t0  = arith.addf %xm, %xp : f64",
               "%t1  = arith.addf %t0, %ym : f64",
               "%t2  = arith.addf %t1, %yp : f64",
               "%t3  = arith.addf %t2, %zm : f64",
               "%t4  = arith.addf %t3, %zp : f64",
               "%t5  = arith.mulf %six, %c : f64",
               "%t6  = arith.subf %t4, %t5 : f64",
               "%lap = arith.mulf %dxinv2, %t6 : f64",
               "neptune_ir.yield %lap : f64"]
        return acc, ops
    if kind == "3d27":
        acc = [("c", (0, 0, 0))]
        names = []
        for di in (-1, 0, 1):
            for dj in (-1, 0, 1):
                for dk in (-1, 0, 1):
                    if (di, dj, dk) == (0, 0, 0):
                        continue
                    nm = "a" + "".join("mzp"[x + 1] for x in (di, dj, dk))
                    acc.append((nm, (di, dj, dk)))
                    names.append(nm)
        ops = ["%c26    = arith.constant 26.0 : f32", "%dxinv2 = arith.constant 0.015625 : f32"]
        prev = names[0]
        for t, nm in enumerate(names[1:]):
            ops.append(f"%s{t} = arith.addf %{prev}, %{nm} : f32")
            prev = f"s{t}"
        ops += ["%t0  = arith.mulf %c26, %c : f32", f"%t1  = arith.subf %{prev}, %t0 : f32",
                "%lap = arith.mulf %dxinv2, %t1 : f32", "neptune_ir.yield %lap : f32"]
        return acc, ops
    raise KeyError(kind)


def stencil_module(kind, shape, origin=None, bounds=None, time_step=None):
    """NeptuneIR module text for fixture `kind` on a field of the given shape.

    origin : logical lower corner of the field box (default all zeros)
    bounds : (lb, ub) of the apply in logical coordinates (default: the interior, one cell in
             from every face)"""
    rank, elem, opname, _, _, title = KINDS[kind]
    shape = [int(x) for x in shape]
    if len(shape) != rank:
        raise ValueError(f"{kind} needs {rank} extents")
    if min(shape) < 3:
        raise ValueError("every extent must be >= 3 (one interior cell)")
    accesses, body_lines = _body(kind)
    origin = [0] * rank if origin is None else [int(x) for x in origin]
    if bounds is None:
        bounds = ([o + 1 for o in origin], [o + n - 1 for o, n in zip(origin, shape)])
    lb0 = ", ".join(str(o) for o in origin)
    ub0 = ", ".join(str(o + n) for o, n in zip(origin, shape))
    lbi = ", ".join(str(int(x)) for x in bounds[0])
    ubi = ", ".join(str(int(x)) for x in bounds[1])
    mr = "x".join(["?"] * rank) + "x" + elem
    idx = ", ".join(f"%i{d}: index" for d in range(rank))
    dims = "x".join(str(n) for n in shape) if len(set(shape)) > 1 else f"{shape[0]}^{rank}"
    if rank == 2 and len(set(shape)) == 1:
        dims = f"{shape[0]}x{shape[0]}"
    out = []
    out.append("// RUN: neptune-opt %s --neptuneir-to-llvm")
    out.append("// " + title.format(dims=dims))
    out.append("// Authored for the MI355X backend: the reference README cites a file of this name")
    out.append("// (README.md:60-64) but ships none.  Structure follows the reference's own smoke inputs")
    out.append("// (test/smoke_tests/smoke_time_advance.mlir:3-31, 53-59, 82-84): interior bounds,")
    out.append("// region signature (index x rank, temp x inputs), body ops in a fixed textual order.")
    out.append("")
    out.append('#loc = #neptune_ir.location<"cell">')
    out.append(f"#b   = #neptune_ir.bounds<lb = [{lb0}], ub = [{ub0}]>")
    out.append("")
    out.append(f"!temp  = !neptune_ir.temp<element = {elem}, bounds = #b, location = #loc>")
    out.append(f"!field = !neptune_ir.field<element = {elem}, bounds = #b, location = #loc>")
    out.append("")
    out.append("module {")
    out.append(f"  neptune_ir.linear_opdef @{opname} : (!temp) -> !temp {{")
    out.append("  ^bb0(%u: !temp):")
    out.append(f"    %r = neptune_ir.apply(%u) attributes {{bounds = #neptune_ir.bounds<lb = [{lbi}], ub = [{ubi}]>}}")
    out.append("      : (!temp) -> !temp {")
    out.append(f"      ^bb0({idx}, %u_in: !temp):")
    for name, off in accesses:
        o = ", ".join(str(x) for x in off)
        out.append(f"        %{name} = neptune_ir.access %u_in[{o}] : !temp -> {elem}")
    out.append("")
    for line in body_lines:
        out.append("        " + line)
    out.append("      }")
    out.append("    neptune_ir.return %r : !temp")
    out.append("  }")
    out.append("")
    out.append(f"  func.func @entry(%out: memref<{mr}>, %in: memref<{mr}>) -> memref<{mr}> {{")
    out.append(f"    %fout = neptune_ir.wrap %out : memref<{mr}> -> !field")
    out.append(f"    %fin  = neptune_ir.wrap %in  : memref<{mr}> -> !field")
    out.append("    %u0   = neptune_ir.load %fin : !field -> !temp")
    out.append(f"    %y    = neptune_ir.apply_linear @{opname}(%u0) : (!temp) -> !temp")
    out.append("    neptune_ir.store %y to %fout : !temp to !field")
    out.append(f"    %res  = neptune_ir.unwrap %fout : !field -> memref<{mr}>")
    out.append(f"    func.return %res : memref<{mr}>")
    out.append("  }")
    if time_step is not None and elem == "f64":
        # explicit Euler step with the operator as rhs: u1 = u0 + dt * A(u0)
        out.append("")
        out.append(f"  func.func @step(%out: memref<{mr}>, %in: memref<{mr}>) -> memref<{mr}> {{")
        out.append(f"    %fout = neptune_ir.wrap %out : memref<{mr}> -> !field")
        out.append(f"    %fin  = neptune_ir.wrap %in  : memref<{mr}> -> !field")
        out.append("    %u0   = neptune_ir.load %fin : !field -> !temp")
        out.append(f"    %dt   = arith.constant {time_step!r} : f64")
        out.append(f"    %u1   = neptune_ir.time_advance %u0, %dt {{method = 0 : i32, rhs = @{opname}}} : !temp, f64 -> !temp")
        out.append("    neptune_ir.store %u1 to %fout : !temp to !field")
        out.append(f"    %res  = neptune_ir.unwrap %fout : !field -> memref<{mr}>")
        out.append(f"    func.return %res : memref<{mr}>")
        out.append("  }")
    out.append("}")
    return "\n".join(out) + "\n"


def main(argv):
    if len(argv) < 3 or argv[1] not in KINDS:
        sys.stderr.write(__doc__)
        return 2
    kind = argv[1]
    rank = KINDS[kind][0]
    time_step = None
    if "--time-step" in argv:
        k = argv.index("--time-step")
        time_step = float(argv[k + 1])
        del argv[k:k + 2]
    dims = [int(x) for x in argv[2:]]
    if len(dims) == 1:
        dims = dims * rank
    sys.stdout.write(stencil_module(kind, dims, time_step=time_step))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))

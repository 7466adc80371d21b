#!/usr/bin/env python3
"""Staggered (MAC) grid: a cell-centred pressure-like field corrected by the divergence of a face-centred velocity,

    p_new = p - (dt / h) * ( u[i+1,j,k] - u[i,j,k] + v[i,j+1,k] - v[i,j,k] + w[i,j,k+1] - w[i,j,k] )

written with the Python DSL as ONE four-input apply.  The three velocity components live on the faces of their own
direction, so each has one more cell along that direction than the result: inputs in boxes of their own -- which the
reference's apply allows (only input 0 must have the result's shape; every input indexes through its own lower bounds,
lib/Passes/DataflowLowering.cpp:283-287, 382-410).  The march kernel reads them through per-input views;
before round 3 such an apply ran on the direct kernel.

usage: examples/staggered_projection.py [N] [REPS]        (default 384^3 cells, 50 launches)"""
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "neptune-pde-solver_amd"))


def build(n, scale):
    import neptune as nep
    nep.reset()
    cells = ([0, 0, 0], [n, n, n])
    faces = [([0, 0, 0], [n + (d == 0), n + (d == 1), n + (d == 2)]) for d in range(3)]
    c = nep.get_compiler()
    c.start_function("project", [("memref", 3)] * 5)
    f_out = nep.wrap(nep.Expr(c.get_function_arg(0)), cells)
    p = nep.load(nep.wrap(nep.Expr(c.get_function_arg(1)), cells))
    u, v, w = (nep.load(nep.wrap(nep.Expr(c.get_function_arg(2 + d)), faces[d], location="face")) for d in range(3))

    @nep.apply(inputs=[p, u, v, w], bounds=cells)
    def corrected(pc, uf, vf, wf):
        div = (uf[1, 0, 0] - uf[0, 0, 0]) + (vf[0, 1, 0] - vf[0, 0, 0]) + (wf[0, 0, 1] - wf[0, 0, 0])
        return pc[0, 0, 0] - scale * div

    nep.store(corrected, f_out)
    c.create_return(nep.unwrap(f_out)._handle)
    c.end_function()
    mod = nep.jit_compile(c)
    nep.reset()
    return mod


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 384
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    import torch
    scale = 0.125
    mod = build(n, scale)
    print("kernels:", [(a["function"], a["kernel"]) for a in mod.report["applies"]])
    gen = torch.Generator(device="cuda").manual_seed(7)
    p = torch.rand((n, n, n), dtype=torch.float64, device="cuda", generator=gen)
    u = torch.rand((n + 1, n, n), dtype=torch.float64, device="cuda", generator=gen)
    v = torch.rand((n, n + 1, n), dtype=torch.float64, device="cuda", generator=gen)
    w = torch.rand((n, n, n + 1), dtype=torch.float64, device="cuda", generator=gen)
    out = torch.zeros_like(p)
    mod.call("project", out, p, u, v, w)
    torch.cuda.synchronize()
    # the same operations in the same order with torch (elementwise IEEE ops: bit-identical)
    div = (u[1:] - u[:-1]) + (v[:, 1:] - v[:, :-1]) + (w[:, :, 1:] - w[:, :, :-1])
    want = p - scale * div
    agree = bool(torch.equal(out, want))
    for _ in range(5):
        mod.call("project", out, p, u, v, w)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        mod.call("project", out, p, u, v, w)
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / reps
    nbytes = (p.numel() + u.numel() + v.numel() + w.numel() + out.numel()) * 8
    print(f"{n}^3 cells, 4 inputs (3 on faces): {per * 1e3:.3f} ms per call, {nbytes / per / 1e12:.2f} TB/s (every field once); "
          f"results agree: {agree}")


if __name__ == "__main__":
    main()

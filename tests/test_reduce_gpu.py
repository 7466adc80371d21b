"""neptune_ir.reduce {kind = "sum"} on the device.

Bar: NOT bit-exact, by construction -- the reference sums serially in row-major order
(lib/Passes/DataflowLowering.cpp:608-638), the device uses a fixed tree.  Stated tolerance, valid
for ANY two summation orders of n terms:  |gpu - serial| <= 2 (n-1) eps sum|x_i|   (each order is
within (n-1) eps sum|x_i| of the exact sum, first order in eps).  The device result must also be
bit-for-bit reproducible from run to run (no atomics), and closer than that bound to the exact sum."""
import math

import numpy as np
import pytest

import helpers
from helpers import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nh(built_libs):
    import torch
    assert torch.cuda.is_available()
    from neptune_hip import _capi, apply, fields
    _capi.load().neptune_hip_init(0)

    class NS:
        pass
    ns = NS()
    ns.capi, ns.apply, ns.fields, ns.torch = _capi, apply, fields, torch
    return ns


def _serial(x):
    dt = x.dtype.type
    return np.cumsum(np.concatenate([np.zeros(1, dt), x.reshape(-1)]), dtype=dt)[-1]


@pytest.mark.parametrize("shape,dtype,origin,bounds", [
    ((1000,), np.float64, (0,), None),
    ((37, 129), np.float64, (3, -2), ((5, 0), (30, 100))),
    ((64, 96, 128), np.float64, (0, 0, 0), None),
    ((64, 96, 128), np.float32, (0, 0, 0), None),
    ((20, 33, 65), np.float32, (1, 1, 1), ((2, 5, 7), (19, 30, 60))),
    ((8, 8), np.float64, (0, 0), ((3, 3), (3, 8))),          # empty domain -> 0
])
def test_reduce_sum_within_stated_tolerance(nh, shape, dtype, origin, bounds):
    x = helpers.hash_field(shape, dtype, seed=13)
    f = nh.fields.DeviceField.from_numpy(x, origin)
    got = nh.apply.reduce_sum(f, bounds)
    sub = x if bounds is None else x[tuple(slice(l - o, u - o) for l, u, o in zip(bounds[0], bounds[1], origin))]
    serial = float(_serial(np.ascontiguousarray(sub)))
    exact = math.fsum(float(v) for v in sub.reshape(-1))
    n = sub.size
    eps = float(np.finfo(dtype).eps)
    bound = 2.0 * max(n - 1, 0) * eps * float(np.abs(sub.astype(np.float64)).sum())
    assert abs(got - serial) <= bound, (got, serial, bound)
    # the tree sum is far more accurate than the worst case: error grows ~log n, not n
    assert abs(got - exact) <= max(64 * eps * float(np.abs(sub.astype(np.float64)).sum()) / max(math.sqrt(n), 1), 4 * eps * abs(exact)) or n == 0
    again = nh.apply.reduce_sum(f, bounds)
    assert again == got                                   # run-to-run reproducible, bit for bit
    if n == 0:
        assert got == 0.0


def test_lowered_norm_module(nh, tmp_path, monkeypatch):
    monkeypatch.setenv("NEPTUNE_CACHE_DIR", str(tmp_path))
    from neptune_hip import lowering
    from test_lowering import NORM
    n0, n1 = 40, 256
    text = NORM.format(n0=n0, n1=n1, m0=n0 - 1, m1=n1 - 1)
    mod = lowering.compile_module(text)
    a = helpers.hash_field((n0, n1), np.float64, seed=6)
    want = float(oracle.Module.parse(text).call("norm2", a))     # serial order
    got_h = mod.call("norm2", a)                                  # host buffer
    got_d = mod.call("norm2", nh.torch.from_numpy(a).cuda())      # device buffer
    assert got_h == got_d
    n = (n0 - 2) * (n1 - 2)
    eps = float(np.finfo(np.float64).eps)
    # sqrt halves the relative error of its argument; bound on the sum as stated above
    ssum = float((a[1:-1, 1:-1] ** 2).sum())
    assert abs(got_h * got_h - want * want) <= 2 * (n - 1) * eps * ssum + 8 * eps * want * want


def _dot_module(shape, elem, stencil, keep_temp, aligned=False):
    """@dot(a, b) -> sum over a sub-box of apply(a, b){a*b (+ a neighbour term)}; keep_temp adds a second use
    of the apply result (a store into a), which forces the unfused apply-then-reduce path"""
    rank = len(shape)
    mr = "x".join("?" * rank) + "x" + elem
    ub = ", ".join(map(str, shape))
    z = ", ".join("0" * rank)
    lbi = ", ".join(["1"] * rank)
    ubi = ", ".join(str(n - 1) for n in shape)
    red_lb = ", ".join(["0"] + ["1"] * (rank - 1))          # plane 0 lies outside apply.bounds: copy-through cells
    red_ub = ", ".join(str(n - (0 if d == 0 else 2)) for d, n in enumerate(shape))
    if aligned:                                             # whole rows: the 16-byte-load kernel can take it
        red_lb = ", ".join(["0"] * rank)
        red_ub = ", ".join(str(n - (1 if (d == 0 and rank > 1) else 0)) for d, n in enumerate(shape))
    idx = ", ".join(f"%i{d}: index" for d in range(rank))
    off = lambda d, o: ", ".join(str(o if k == d else 0) for k in range(rank))
    body = [f"        %x = neptune_ir.access %p[{z}] : !t -> {elem}",
            f"        %y = neptune_ir.access %q[{z}] : !t -> {elem}",
            f"        %m = arith.mulf %x, %y : {elem}"]
    if stencil:
        body += [f"        %l = neptune_ir.access %p[{off(rank - 1, -1)}] : !t -> {elem}",
                 f"        %r = neptune_ir.access %q[{off(0, 1)}] : !t -> {elem}",
                 f"        %lr = arith.subf %l, %r : {elem}",
                 f"        %v = arith.addf %m, %lr : {elem}"]
    else:
        body += [f"        %v = arith.addf %m, %m : {elem}"]
    keep = "    neptune_ir.store %w to %fa : !t to !f\n" if keep_temp else ""
    return f"""
#l = #neptune_ir.location<"cell">
#b = #neptune_ir.bounds<lb = [{z}], ub = [{ub}]>
!t = !neptune_ir.temp<element = {elem}, bounds = #b, location = #l>
!f = !neptune_ir.field<element = {elem}, bounds = #b, location = #l>
module {{
  func.func @dot(%a: memref<{mr}>, %b: memref<{mr}>) -> {elem} {{
    %fa = neptune_ir.wrap %a : memref<{mr}> -> !f
    %fb = neptune_ir.wrap %b : memref<{mr}> -> !f
    %u = neptune_ir.load %fa : !f -> !t
    %v = neptune_ir.load %fb : !f -> !t
    %w = neptune_ir.apply(%u, %v) attributes {{bounds = #neptune_ir.bounds<lb = [{lbi}], ub = [{ubi}]>}} : (!t, !t) -> !t {{
      ^bb0({idx}, %p: !t, %q: !t):
{chr(10).join(body)}
        neptune_ir.yield %v : {elem}
    }}
    %s = neptune_ir.reduce %w in #neptune_ir.bounds<lb = [{red_lb}], ub = [{red_ub}]> {{kind = "sum"}} : !t -> {elem}
{keep}    func.return %s : {elem}
  }}
}}
"""


@pytest.mark.parametrize("shape,elem,stencil,aligned", [
    ((9, 7, 300), "f64", True, False), ((5, 6, 128), "f32", False, False), ((33, 5000), "f64", True, False),
    ((70000,), "f64", False, False), ((4100,), "f32", True, False), ((3, 3, 5), "f64", True, False),
    # pointwise + whole aligned rows: the 16-byte-load kernel (rim cells of every row are copy-through cells)
    ((6, 5, 128), "f32", False, True), ((33, 4096), "f64", False, True), ((70000,), "f64", False, True),
    ((7, 9, 1100), "f64", False, True), ((9, 7, 300), "f64", True, True)])
def test_fused_apply_reduce_dot_products(nh, tmp_path, monkeypatch, shape, elem, stencil, aligned):
    """reduce(apply(...)) with a single-use apply result runs as ONE kernel (no temp, nothing written): same
    semantics as the two ops back to back, incl. copy-through cells inside the reduced box; within the stated
    tolerance of the oracle's serial sum and of the unfused path"""
    monkeypatch.setenv("NEPTUNE_CACHE_DIR", str(tmp_path))
    from neptune_hip import lowering
    dt = np.float64 if elem == "f64" else np.float32
    a = helpers.hash_field(shape, dt, seed=31)
    b = helpers.hash_field(shape, dt, seed=32)
    fused_text = _dot_module(shape, elem, stencil, keep_temp=False, aligned=aligned)
    plain_text = _dot_module(shape, elem, stencil, keep_temp=True, aligned=aligned)
    fused, plain = lowering.compile_module(fused_text), lowering.compile_module(plain_text)
    assert [x["kernel"] for x in fused.report["applies"]] == ["reduce"]
    assert [x["kernel"] for x in plain.report["applies"]] != ["reduce"]
    want = float(oracle.Module.parse(fused_text).call("dot", a.copy(), b.copy()))
    got = fused.call("dot", nh.torch.from_numpy(a).cuda(), nh.torch.from_numpy(b).cuda())
    assert got == fused.call("dot", a.copy(), b.copy())                 # host buffers, and reproducible
    unfused = plain.call("dot", nh.torch.from_numpy(a).cuda(), nh.torch.from_numpy(b).cuda())
    # bound: both orders are within (n-1) eps sum|x_i| of the exact sum; |x_i| <= 3 here
    n = int(np.prod(shape))
    tol = 2 * (n - 1) * float(np.finfo(dt).eps) * 3.0 * n
    assert abs(got - want) <= tol and abs(unfused - want) <= tol
    # and tight in practice: the tree sums are far more accurate than the bound
    ref = abs(want) + 1.0
    assert abs(got - want) <= (1e-9 if elem == "f64" else 2e-3) * ref


def test_reduce_rejects_bad_domains(nh):
    f = nh.fields.DeviceField.from_numpy(np.ones((4, 4)))
    with pytest.raises(nh.capi.NeptuneHipError, match="EOOB"):
        nh.apply.reduce_sum(f, ((0, 0), (5, 4)))

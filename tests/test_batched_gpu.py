"""Applies of rank 4..6 (leading batch / component dimensions; the reference's apply lowering is rank-generic,
lib/Passes/DataflowLowering.cpp:268-270, 301-308): the HIP lowering peels the leading dimensions off and launches one rank-3
apply per leading index (lowered_runtime.hpp run_apply_batched).  Bit for bit against the oracle, host and device arguments,
in place, several tiles; a whole-buffer reduce over the rank-4 result within the stated tolerance of tests/test_reduce_gpu.py."""
import os
from pathlib import Path

import numpy as np
import pytest

import helpers
from helpers import bits_equal, mismatch_report, oracle

FIXTURE = Path(__file__).resolve().parent / "mlir_tests" / "conversion_tests" / "apply-4d-batch-7pt.mlir"


def rank5_text():
    """(2, 3, z, y, x) f32, two inputs (one read at offsets incl. a radius-2 tap, one at the centre), both leading indices in
    the body, bounds cutting into both leading dimensions"""
    return """
#l = #neptune_ir.location<"cell">
#b = #neptune_ir.bounds<lb = [-1, 2, 0, 0, 0], ub = [1, 5, 9, 11, 260]>
!t = !neptune_ir.temp<element = f32, bounds = #b, location = #l>
!f = !neptune_ir.field<element = f32, bounds = #b, location = #l>
module {
  neptune_ir.nonlinear_opdef @op : (!t, !t) -> !t {
  ^bb0(%u: !t, %w: !t):
    %r = neptune_ir.apply(%u, %w) attributes {bounds = #neptune_ir.bounds<lb = [0, 2, 2, 1, 2], ub = [1, 4, 7, 10, 258]>} : (!t, !t) -> !t {
      ^bb0(%p: index, %q: index, %i: index, %j: index, %k: index, %a: !t, %c: !t):
        %v0 = neptune_ir.access %a[0, 0, 0, 0, 0] : !t -> f32
        %v1 = neptune_ir.access %a[0, 0, -2, 0, 0] : !t -> f32
        %v2 = neptune_ir.access %a[0, 0, 1, 0, 0] : !t -> f32
        %v3 = neptune_ir.access %a[0, 0, 0, -1, 0] : !t -> f32
        %v4 = neptune_ir.access %a[0, 0, 0, 0, 2] : !t -> f32
        %v5 = neptune_ir.access %c[0, 0, 0, 0, 0] : !t -> f32
        %s0 = arith.addf %v1, %v2 : f32
        %s1 = arith.subf %s0, %v3 : f32
        %s2 = arith.addf %s1, %v4 : f32
        %pi = arith.index_cast %p : index to i64
        %pf = arith.sitofp %pi : i64 to f32
        %qi = arith.index_cast %q : index to i64
        %qf = arith.sitofp %qi : i64 to f32
        %ki = arith.index_cast %k : index to i64
        %kf = arith.sitofp %ki : i64 to f32
        %m0 = arith.mulf %qf, %v5 : f32
        %m1 = arith.mulf %pf, %v0 : f32
        %s3 = arith.addf %s2, %m0 : f32
        %s4 = arith.subf %s3, %m1 : f32
        %o = arith.addf %s4, %kf : f32
        neptune_ir.yield %o : f32
    }
    neptune_ir.return %r : !t
  }
  func.func @entry(%out: memref<?x?x?x?x?xf32>, %in0: memref<?x?x?x?x?xf32>, %in1: memref<?x?x?x?x?xf32>) -> memref<?x?x?x?x?xf32> {
    %fo = neptune_ir.wrap %out : memref<?x?x?x?x?xf32> -> !f
    %f0 = neptune_ir.wrap %in0 : memref<?x?x?x?x?xf32> -> !f
    %f1 = neptune_ir.wrap %in1 : memref<?x?x?x?x?xf32> -> !f
    %t0 = neptune_ir.load %f0 : !f -> !t
    %t1 = neptune_ir.load %f1 : !f -> !t
    %y = neptune_ir.apply_nonlinear @op(%t0, %t1) : (!t, !t) -> !t
    neptune_ir.store %y to %fo : !t to !f
    %res = neptune_ir.unwrap %fo : !f -> memref<?x?x?x?x?xf32>
    func.return %res : memref<?x?x?x?x?xf32>
  }
  func.func @boxed(%out: memref<?x?x?x?x?xf32>, %in0: memref<?x?x?x?x?xf32>, %in1: memref<?x?x?x?x?xf32>) -> f32 {
    %fo = neptune_ir.wrap %out : memref<?x?x?x?x?xf32> -> !f
    %f0 = neptune_ir.wrap %in0 : memref<?x?x?x?x?xf32> -> !f
    %f1 = neptune_ir.wrap %in1 : memref<?x?x?x?x?xf32> -> !f
    %t0 = neptune_ir.load %f0 : !f -> !t
    %t1 = neptune_ir.load %f1 : !f -> !t
    %y = neptune_ir.apply_nonlinear @op(%t0, %t1) : (!t, !t) -> !t
    neptune_ir.store %y to %fo {bounds = #neptune_ir.bounds<lb = [-1, 3, 1, 0, 5], ub = [1, 5, 8, 11, 250]>} : !t to !f
    %s = neptune_ir.reduce %t1 in #neptune_ir.bounds<lb = [0, 2, 2, 2, 3], ub = [1, 4, 9, 9, 257]> {kind = "sum"} : !t -> f32
    func.return %s : f32
  }
  func.func @inplace(%in0: memref<?x?x?x?x?xf32>, %in1: memref<?x?x?x?x?xf32>) -> memref<?x?x?x?x?xf32> {
    %f0 = neptune_ir.wrap %in0 : memref<?x?x?x?x?xf32> -> !f
    %f1 = neptune_ir.wrap %in1 : memref<?x?x?x?x?xf32> -> !f
    %t0 = neptune_ir.load %f0 : !f -> !t
    %t1 = neptune_ir.load %f1 : !f -> !t
    %y = neptune_ir.apply_nonlinear @op(%t0, %t1) : (!t, !t) -> !t
    neptune_ir.store %y to %f0 : !t to !f
    %res = neptune_ir.unwrap %f0 : !f -> memref<?x?x?x?x?xf32>
    func.return %res : memref<?x?x?x?x?xf32>
  }
}
"""


def test_rank4_and_rank5_applies_are_lowered_with_their_leading_dimensions_peeled_off():
    from neptune_hip import lowering
    src, rep = lowering.to_hip(FIXTURE.read_text())
    assert rep["lowered"] == ["lapc", "entry", "norm2"] and not rep.get("skipped")
    a = rep["applies"][0]
    assert a["rank"] == 4 and a["kernel"] == "march" and a["geom_symbol"] == ""     # no geometry-level entry beyond rank 3
    assert "nl::run_apply_batched<Body_lapc_0, double, 4, 1, FP_lapc_0>" in src
    assert "a.template get<0, -1, 0, 0>()" in src and "(int64_t)lead[0]" in src     # kernel offsets: the last three dimensions
    assert "neptune_hip::Footprint<0, 1, 1, 1, false, true>" in src
    src5, rep5 = lowering.to_hip(rank5_text())
    assert rep5["lowered"] == ["op", "entry", "boxed", "inplace"]
    assert "lead[0]" in src5 and "lead[1]" in src5 and "a.template idx<2>()" in src5
    # an offset along a leading dimension makes it a stencil in four dimensions: the rank-generic kernel, every offset kept
    nd = FIXTURE.read_text().replace("%a[0, -1, 0, 0]", "%a[-1, 0, 0, 0]")
    srcn, repn = lowering.to_hip(nd)
    assert repn["lowered"] == ["lapc", "entry", "norm2"]
    assert repn["applies"][0]["rank"] == 4 and repn["applies"][0]["kernel"] == "direct" and repn["applies"][0]["geom_symbol"] == ""
    assert "nl::run_apply_nd<Body_lapc_0, double, 4, 1>" in srcn and "a.template get<0, -1, 0, 0, 0>()" in srcn
    assert "lead[" not in srcn and "neptune_hip::ReachN kNdReach_lapc_0" in srcn


def stencil4d_text(elem="f64"):
    """(t, z, y, x): a 9-point star in FOUR dimensions on input 0 plus a second input in a box of its own (one more cell on
    either side of dim 0 and of dim 3) read at offsets along dims 0 and 3, every index argument in the body, bounds that cut
    into every dimension"""
    return f"""
#l = #neptune_ir.location<"cell">
#b = #neptune_ir.bounds<lb = [2, 0, -1, 0], ub = [9, 6, 8, 140]>
#bw = #neptune_ir.bounds<lb = [1, 0, -1, -1], ub = [10, 6, 8, 141]>
!t = !neptune_ir.temp<element = {elem}, bounds = #b, location = #l>
!f = !neptune_ir.field<element = {elem}, bounds = #b, location = #l>
!tw = !neptune_ir.temp<element = {elem}, bounds = #bw, location = #l>
!fw = !neptune_ir.field<element = {elem}, bounds = #bw, location = #l>
module {{
  neptune_ir.nonlinear_opdef @op : (!t, !tw) -> !t {{
  ^bb0(%u: !t, %w: !tw):
    %r = neptune_ir.apply(%u, %w) attributes {{bounds = #neptune_ir.bounds<lb = [3, 1, 0, 1], ub = [8, 5, 7, 139]>}} : (!t, !tw) -> !t {{
      ^bb0(%p: index, %i: index, %j: index, %k: index, %a: !t, %c: !tw):
        %v0 = neptune_ir.access %a[0, 0, 0, 0] : !t -> {elem}
        %v1 = neptune_ir.access %a[-1, 0, 0, 0] : !t -> {elem}
        %v2 = neptune_ir.access %a[1, 0, 0, 0] : !t -> {elem}
        %v3 = neptune_ir.access %a[0, -1, 0, 0] : !t -> {elem}
        %v4 = neptune_ir.access %a[0, 1, 0, 0] : !t -> {elem}
        %v5 = neptune_ir.access %a[0, 0, -1, 0] : !t -> {elem}
        %v6 = neptune_ir.access %a[0, 0, 1, 0] : !t -> {elem}
        %v7 = neptune_ir.access %a[0, 0, 0, -1] : !t -> {elem}
        %v8 = neptune_ir.access %a[0, 0, 0, 1] : !t -> {elem}
        %w0 = neptune_ir.access %c[-2, 0, 0, 0] : !tw -> {elem}
        %w1 = neptune_ir.access %c[2, 0, 0, -2] : !tw -> {elem}
        %w2 = neptune_ir.access %c[1, 0, 0, 2] : !tw -> {elem}
        %c8 = arith.constant 8.0 : {elem}
        %c3 = arith.constant 0.375 : {elem}
        %s0 = arith.addf %v1, %v2 : {elem}
        %s1 = arith.addf %s0, %v3 : {elem}
        %s2 = arith.addf %s1, %v4 : {elem}
        %s3 = arith.addf %s2, %v5 : {elem}
        %s4 = arith.addf %s3, %v6 : {elem}
        %s5 = arith.addf %s4, %v7 : {elem}
        %s6 = arith.addf %s5, %v8 : {elem}
        %m0 = arith.mulf %c8, %v0 : {elem}
        %s7 = arith.subf %s6, %m0 : {elem}
        %x0 = arith.subf %w0, %w1 : {elem}
        %x1 = arith.mulf %c3, %x0 : {elem}
        %x2 = arith.addf %x1, %w2 : {elem}
        %s8 = arith.addf %s7, %x2 : {elem}
        %pi = arith.index_cast %p : index to i64
        %pf = arith.sitofp %pi : i64 to {elem}
        %ki = arith.index_cast %k : index to i64
        %kf = arith.sitofp %ki : i64 to {elem}
        %ji = arith.index_cast %j : index to i64
        %jf = arith.sitofp %ji : i64 to {elem}
        %y0 = arith.mulf %pf, %kf : {elem}
        %y1 = arith.subf %y0, %jf : {elem}
        %o = arith.addf %s8, %y1 : {elem}
        neptune_ir.yield %o : {elem}
    }}
    neptune_ir.return %r : !t
  }}
  func.func @entry(%out: memref<?x?x?x?x{elem}>, %in0: memref<?x?x?x?x{elem}>, %in1: memref<?x?x?x?x{elem}>) -> memref<?x?x?x?x{elem}> {{
    %fo = neptune_ir.wrap %out : memref<?x?x?x?x{elem}> -> !f
    %f0 = neptune_ir.wrap %in0 : memref<?x?x?x?x{elem}> -> !f
    %f1 = neptune_ir.wrap %in1 : memref<?x?x?x?x{elem}> -> !fw
    %t0 = neptune_ir.load %f0 : !f -> !t
    %t1 = neptune_ir.load %f1 : !fw -> !tw
    %y = neptune_ir.apply_nonlinear @op(%t0, %t1) : (!t, !tw) -> !t
    neptune_ir.store %y to %fo : !t to !f
    %res = neptune_ir.unwrap %fo : !f -> memref<?x?x?x?x{elem}>
    func.return %res : memref<?x?x?x?x{elem}>
  }}
  func.func @inplace(%in0: memref<?x?x?x?x{elem}>, %in1: memref<?x?x?x?x{elem}>) -> memref<?x?x?x?x{elem}> {{
    %f0 = neptune_ir.wrap %in0 : memref<?x?x?x?x{elem}> -> !f
    %f1 = neptune_ir.wrap %in1 : memref<?x?x?x?x{elem}> -> !fw
    %t0 = neptune_ir.load %f0 : !f -> !t
    %t1 = neptune_ir.load %f1 : !fw -> !tw
    %y = neptune_ir.apply_nonlinear @op(%t0, %t1) : (!t, !tw) -> !t
    neptune_ir.store %y to %f0 : !t to !f
    %res = neptune_ir.unwrap %f0 : !f -> memref<?x?x?x?x{elem}>
    func.return %res : memref<?x?x?x?x{elem}>
  }}
}}
"""


def stencil5d_text():
    """rank 5, f32, offsets along dims 0, 1 and 4 of one input; an scf.if on a leading index"""
    return """
#l = #neptune_ir.location<"cell">
#b = #neptune_ir.bounds<lb = [0, 0, 0, 0, 0], ub = [4, 5, 3, 6, 70]>
!t = !neptune_ir.temp<element = f32, bounds = #b, location = #l>
!f = !neptune_ir.field<element = f32, bounds = #b, location = #l>
module {
  neptune_ir.nonlinear_opdef @op : (!t) -> !t {
  ^bb0(%u: !t):
    %r = neptune_ir.apply(%u) attributes {bounds = #neptune_ir.bounds<lb = [1, 1, 0, 0, 1], ub = [3, 4, 3, 6, 69]>} : (!t) -> !t {
      ^bb0(%p: index, %q: index, %i: index, %j: index, %k: index, %a: !t):
        %v0 = neptune_ir.access %a[0, 0, 0, 0, 0] : !t -> f32
        %v1 = neptune_ir.access %a[-1, 0, 0, 0, 0] : !t -> f32
        %v2 = neptune_ir.access %a[1, 1, 0, 0, 0] : !t -> f32
        %v3 = neptune_ir.access %a[0, -1, 0, 0, 1] : !t -> f32
        %v4 = neptune_ir.access %a[0, 0, 0, 0, -1] : !t -> f32
        %c1 = arith.constant 1 : index
        %is1 = arith.cmpi eq, %p, %c1 : index
        %s0 = arith.addf %v1, %v2 : f32
        %s1 = arith.subf %s0, %v3 : f32
        %e = scf.if %is1 -> (f32) {
          %h = arith.addf %s1, %v4 : f32
          scf.yield %h : f32
        } else {
          %h2 = arith.mulf %s1, %v0 : f32
          scf.yield %h2 : f32
        }
        neptune_ir.yield %e : f32
    }
    neptune_ir.return %r : !t
  }
  func.func @entry(%out: memref<?x?x?x?x?xf32>, %in0: memref<?x?x?x?x?xf32>) -> memref<?x?x?x?x?xf32> {
    %fo = neptune_ir.wrap %out : memref<?x?x?x?x?xf32> -> !f
    %f0 = neptune_ir.wrap %in0 : memref<?x?x?x?x?xf32> -> !f
    %t0 = neptune_ir.load %f0 : !f -> !t
    %y = neptune_ir.apply_nonlinear @op(%t0) : (!t) -> !t
    neptune_ir.store %y to %fo : !t to !f
    %res = neptune_ir.unwrap %fo : !f -> memref<?x?x?x?x?xf32>
    func.return %res : memref<?x?x?x?x?xf32>
  }
}
"""


def test_stencils_in_more_than_three_dimensions_are_lowered_onto_the_rank_generic_kernel():
    from neptune_hip import lowering
    src, rep = lowering.to_hip(stencil4d_text())
    assert rep["lowered"] == ["op", "entry", "inplace"] and not rep.get("skipped")
    assert "nl::run_apply_nd<Body_op_0, double, 4, 2>" in src and "a.template get<1, 2, 0, 0, -2>()" in src
    assert "a.template idx<0>()" in src and "a.template idx<3>()" in src and "lead[" not in src
    # reach of the unconditional accesses per input and dimension: {lo}, {hi}
    assert "kNdReach_op_0 = {{{-1, -1, -1, -1, 1, 1}, {-2, 0, 0, -2, 1, 1}, " in src
    assert "{{1, 1, 1, 1, -1, -1}, {2, 0, 0, 2, -1, -1}, " in src
    src5, rep5 = lowering.to_hip(stencil5d_text())
    assert rep5["lowered"] == ["op", "entry"] and "nl::run_apply_nd<Body_op_0, float, 5, 1>" in src5


@pytest.fixture(scope="module")
def env(built_libs, tmp_path_factory):
    import torch
    assert torch.cuda.is_available()
    os.environ["NEPTUNE_CACHE_DIR"] = str(tmp_path_factory.mktemp("neptune_cache_batched"))
    from neptune_hip import lowering
    return lowering, torch


@pytest.mark.gpu
def test_rank4_apply_matches_the_oracle(env, monkeypatch):
    lowering, torch = env
    text = FIXTURE.read_text()
    shape = (3, 10, 12, 256)
    u = helpers.hash_field(shape, np.float64, seed=5)
    want = np.full(shape, -3.0)
    m = oracle.Module.parse(text)
    m.call("entry", want, u)
    assert bits_equal(want[0], u[0]) and not bits_equal(want[1], u[1])       # component 0: copy-through
    mod = lowering.compile_module(text)
    for setting in ({}, {"NEPTUNE_HIP_VARIANT": "0", "NEPTUNE_HIP_CHUNK": "3"}, {"NEPTUNE_HIP_VARIANT": "7"}, {"NEPTUNE_HIP_KERNEL": "direct"}):
        for k in ("NEPTUNE_HIP_KERNEL", "NEPTUNE_HIP_VARIANT", "NEPTUNE_HIP_CHUNK"):
            monkeypatch.delenv(k, raising=False)
        for k, v in setting.items():
            monkeypatch.setenv(k, v)
        d_out = torch.full(shape, -3.0, dtype=torch.float64, device="cuda")
        mod.call("entry", d_out, torch.from_numpy(u).cuda())
        got = d_out.cpu().numpy()
        assert bits_equal(got, want), f"{setting}: " + mismatch_report(got.reshape(-1, 12, 256), want.reshape(-1, 12, 256))
        h_out = np.full(shape, -3.0)
        res = mod.call("entry", h_out, u)                                    # host arrays: staged, result in the argument
        assert bits_equal(h_out, want) and bits_equal(np.asarray(res), want)
    for k in ("NEPTUNE_HIP_KERNEL", "NEPTUNE_HIP_VARIANT", "NEPTUNE_HIP_CHUNK"):
        monkeypatch.delenv(k, raising=False)
    # whole-buffer reduce over the rank-4 result: fixed tree over the flat buffer, tolerance of tests/test_reduce_gpu.py
    got_sum = mod.call("norm2", u)
    ref = want.astype(np.float64)
    tol = 2 * (ref.size - 1) * np.finfo(np.float64).eps * np.abs(ref).sum()
    assert abs(got_sum - float(np.sum(ref))) <= tol


@pytest.mark.gpu
def test_rank5_apply_two_inputs_in_place(env):
    lowering, torch = env
    text = rank5_text()
    shape = (2, 3, 9, 11, 260)
    a = helpers.hash_field(shape, np.float32, seed=11)
    c = helpers.hash_field(shape, np.float32, seed=12)
    m = oracle.Module.parse(text)
    want = np.full(shape, 9.0, dtype=np.float32)
    m.call("entry", want, a, c)
    mod = lowering.compile_module(text)
    d_out = torch.full(shape, 9.0, dtype=torch.float32, device="cuda")
    mod.call("entry", d_out, torch.from_numpy(a).cuda(), torch.from_numpy(c).cuda())
    got = d_out.cpu().numpy()
    assert bits_equal(got, want), mismatch_report(got.reshape(-1, 11, 260), want.reshape(-1, 11, 260))
    # store apply(load f) to f: the result is a fresh buffer, the field is overwritten afterwards
    a_dev = torch.from_numpy(a).cuda()
    mod.call("inplace", a_dev, torch.from_numpy(c).cuda())
    assert bits_equal(a_dev.cpu().numpy(), want)
    a_host = a.copy()
    mod.call("inplace", a_host, c)
    assert bits_equal(a_host, want)
    # store {bounds} and reduce {bounds} beyond rank 3: one rank-3 box copy / box sum per leading index
    want_b = np.full(shape, 9.0, dtype=np.float32)
    want_s = m.call("boxed", want_b, a, c)
    d_out = torch.full(shape, 9.0, dtype=torch.float32, device="cuda")
    got_s = mod.call("boxed", d_out, torch.from_numpy(a).cuda(), torch.from_numpy(c).cuda())
    assert bits_equal(d_out.cpu().numpy(), want_b)
    assert bits_equal(want_b[0, 0], np.full(shape[2:], 9.0, dtype=np.float32)) and not bits_equal(want_b[1, 2], want_b[0, 0])
    box = c[1:2, 0:2, 2:9, 2:9, 3:257].astype(np.float64)
    tol = 2 * (box.size - 1) * np.finfo(np.float32).eps * np.abs(box).sum()
    assert abs(float(got_s) - float(box.sum())) <= tol and abs(float(want_s) - float(box.sum())) <= tol


@pytest.mark.gpu
@pytest.mark.parametrize("elem", ["f64", "f32"])
def test_a_stencil_in_four_dimensions_matches_the_oracle(env, elem):
    """offsets along ALL four dimensions, a second input in a box of its own: device and host arguments, in place"""
    lowering, torch = env
    dt = np.float64 if elem == "f64" else np.float32
    text = stencil4d_text(elem)
    shape, wshape = (7, 6, 9, 140), (9, 6, 9, 142)
    a = helpers.hash_field(shape, dt, seed=21)
    c = helpers.hash_field(wshape, dt, seed=22)
    m = oracle.Module.parse(text)
    want = np.full(shape, -5.0, dtype=dt)
    m.call("entry", want, a, c)
    assert bits_equal(want[0], a[0]) and not bits_equal(want[2], a[2])        # t = 2 lies outside apply.bounds: copy-through
    mod = lowering.compile_module(text)
    d_out = torch.full(shape, -5.0, dtype=getattr(torch, "float64" if elem == "f64" else "float32"), device="cuda")
    mod.call("entry", d_out, torch.from_numpy(a).cuda(), torch.from_numpy(c).cuda())
    got = d_out.cpu().numpy()
    assert bits_equal(got, want), mismatch_report(got.reshape(-1, 9, 140), want.reshape(-1, 9, 140))
    h_out = np.full(shape, -5.0, dtype=dt)
    mod.call("entry", h_out, a, c)
    assert bits_equal(h_out, want)
    a_dev = torch.from_numpy(a).cuda()
    mod.call("inplace", a_dev, torch.from_numpy(c).cuda())
    assert bits_equal(a_dev.cpu().numpy(), want)


@pytest.mark.gpu
def test_a_stencil_in_five_dimensions_and_a_read_outside_an_input_is_refused(env, tmp_path):
    lowering, torch = env
    text = stencil5d_text()
    shape = (4, 5, 3, 6, 70)
    a = helpers.hash_field(shape, np.float32, seed=31)
    want = np.full(shape, 2.0, dtype=np.float32)
    oracle.Module.parse(text).call("entry", want, a)
    mod = lowering.compile_module(text)
    d_out = torch.full(shape, 2.0, dtype=torch.float32, device="cuda")
    mod.call("entry", d_out, torch.from_numpy(a).cuda())
    got = d_out.cpu().numpy()
    assert bits_equal(got, want), mismatch_report(got.reshape(-1, 6, 70), want.reshape(-1, 6, 70))
    # bounds that let the [-1, ...] access leave the input along a LEADING dimension: refused at run time like rank <= 3
    bad = text.replace("lb = [1, 1, 0, 0, 1], ub = [3, 4, 3, 6, 69]", "lb = [0, 1, 0, 0, 1], ub = [3, 4, 3, 6, 69]")
    import subprocess
    import sys
    script = tmp_path / "oob.py"
    script.write_text(
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {str(Path(lowering.__file__).resolve().parent.parent)!r})\n"
        "from neptune_hip import lowering\n"
        f"mod = lowering.compile_module({bad!r})\n"
        f"a = torch.zeros({shape!r}, dtype=torch.float32, device='cuda')\n"
        "mod.call('entry', torch.zeros_like(a), a)\n")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "reads outside an input's bounds" in r.stderr


STEP4 = """
  func.func @step(%out: memref<?x?x?x?xf64>, %in: memref<?x?x?x?xf64>) -> memref<?x?x?x?xf64> {
    %fo = neptune_ir.wrap %out : memref<?x?x?x?xf64> -> !f
    %fi = neptune_ir.wrap %in : memref<?x?x?x?xf64> -> !f
    %u0 = neptune_ir.load %fi : !f -> !t
    %dt = arith.constant 6.25e-2 : f64
    %u1 = neptune_ir.time_advance %u0, %dt {method = 0 : i32, rhs = @RHS} : !t, f64 -> !t
    %u2 = neptune_ir.time_advance %u1, %dt {method = 0 : i32, rhs = @RHS} : !t, f64 -> !t
    neptune_ir.store %u2 to %fo : !t to !f
    %res = neptune_ir.unwrap %fo : !f -> memref<?x?x?x?xf64>
    func.return %res : memref<?x?x?x?xf64>
  }
}
"""


def step4_texts():
    """explicit time_advance on rank-4 fields: the rhs a batched 3-D operator (leading dimension peeled off) and a stencil in
    four dimensions (rank-generic kernel); two steps chained inside one function"""
    batched = FIXTURE.read_text()
    batched = batched[:batched.rindex("}")] + STEP4.replace("@RHS", "@lapc")
    nd = FIXTURE.read_text().replace("%a[0, -1, 0, 0]", "%a[-1, 0, 0, 0]")
    nd = nd[:nd.rindex("}")] + STEP4.replace("@RHS", "@lapc")
    return {"batched": batched, "nd": nd}


def test_explicit_time_advance_beyond_rank_3_is_lowered():
    from neptune_hip import lowering
    for name, text in step4_texts().items():
        src, rep = lowering.to_hip(text)
        assert "step" in rep["lowered"] and not rep.get("skipped"), name
        assert src.count("nl::run_euler_axpy_flat<double>(sc, ") == 2 and "lapc__impl(sc, " in src


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["batched", "nd"])
def test_explicit_time_advance_on_rank4_fields_matches_the_oracle(env, name):
    lowering, torch = env
    text = step4_texts()[name]
    shape = (3, 10, 12, 256)
    u = helpers.hash_field(shape, np.float64, seed=41)
    want = np.full(shape, -1.0)
    oracle.Module.parse(text).call("step", want, u)
    mod = lowering.compile_module(text)
    d_out = torch.full(shape, -1.0, dtype=torch.float64, device="cuda")
    mod.call("step", d_out, torch.from_numpy(u).cuda())
    got = d_out.cpu().numpy()
    assert bits_equal(got, want), mismatch_report(got.reshape(-1, 12, 256), want.reshape(-1, 12, 256))
    h_out = np.full(shape, -1.0)
    mod.call("step", h_out, u)
    assert bits_equal(h_out, want)

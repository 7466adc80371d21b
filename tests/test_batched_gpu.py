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
    # an offset along a leading dimension is refused with a diagnostic (the function is not lowered)
    bad = FIXTURE.read_text().replace("%a[0, -1, 0, 0]", "%a[-1, 0, 0, 0]")
    with pytest.raises(lowering.LoweringError, match="offset along a leading dimension of a rank-4 apply"):
        lowering.to_hip(bad)


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

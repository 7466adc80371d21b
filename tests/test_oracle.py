"""Pin the oracle (test infrastructure): known-answer vectors, the reference's own input files
(when the reference tree is present), C restatement vs numpy restatement, and its guard rails."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

import helpers
from helpers import bits_equal, mismatch_report, oracle

REFERENCE = Path("/root/reference/test/smoke_tests")


def _vec(hexes):
    return np.array([float.fromhex(h) for h in hexes], dtype=np.float64)


@pytest.fixture(scope="module")
def kat_doc():
    return helpers.load_kats()


@pytest.fixture(scope="module")
def kat_module(kat_doc):
    return oracle.Module.parse((helpers.REPO / kat_doc["ir"]).read_text())


def test_oracle_reproduces_every_known_answer_vector(kat_doc, kat_module):
    assert len(kat_doc["kats"]) >= 6
    for k in kat_doc["kats"]:
        ins = [_vec(v) for v in k["inputs"]]
        want = _vec(k["expected"])
        got = kat_module.call(k["symbol"], *ins)
        assert bits_equal(got, want), k["name"] + "\n" + mismatch_report(got, want)


def test_known_answers_spelled_out():
    """the values SURVEY.md 8c derives by hand, independent of the JSON"""
    m = oracle.Module.parse((helpers.GOLDEN_DIR / "kat_smoke_1d.mlir").read_text())
    u = np.arange(1, 17, dtype=np.float64)
    lap = m.call("kat_lap", u)
    assert lap[0] == 1.0 and lap[15] == 16.0 and np.all(lap[1:15] == 0.0)
    react = m.call("kat_react", u)
    assert react[0] == 1.0 and react[15] == 16.0
    assert react[1] == 2.0 + 1.0e-2 * (2.0 - (2.0 * 2.0) * 2.0)
    F = m.call("kat_resid", u, 0.5 * u)
    assert F[0] == 0.5 and F[15] == 8.0


@pytest.mark.skipif(not REFERENCE.exists(), reason="reference tree not mounted (GPU box)")
def test_oracle_on_the_reference_input_files_themselves(kat_doc):
    """same vectors, but parsed from the reference's own smoke inputs: pins the oracle's parser to
    the reference's textual format (aliases, comments, opdef/func syntax, time_advance skipped)"""
    by_symbol = {}
    for k in kat_doc["kats"]:
        by_symbol.setdefault(k["symbol"], []).append(k)
    for ksym, (fname, refsym) in kat_doc["reference_symbol_of"].items():
        m = oracle.Module.parse((REFERENCE / fname).read_text())
        for k in by_symbol[ksym]:
            got = m.call(refsym, *[_vec(v) for v in k["inputs"]])
            assert bits_equal(got, _vec(k["expected"])), f"{fname} @{refsym}: {k['name']}"
        # the solver/time-stepping ops around them are outside the hot path
        with pytest.raises(oracle.Unsupported):
            n = 32 if "bs" in fname else 16
            m.call("entry", np.zeros(n), np.arange(1, n + 1, dtype=np.float64))


@pytest.mark.skipif(not REFERENCE.exists(), reason="reference tree not mounted (GPU box)")
def test_old_style_region_is_rejected_like_the_reference_verifier_does():
    # smoke_apply.mlir uses ^bb0(%i0: index) capturing the outer temp: fails ApplyOp::verify
    # (lib/Dialect/NeptuneIR/NeptuneIRVerifier.cpp:150-168)
    m = oracle.Module.parse((REFERENCE / "smoke_apply.mlir").read_text())
    with pytest.raises(oracle.OracleError, match="block arg count"):
        m.call("A", np.arange(1, 5, dtype=np.float64))


def test_out_of_bounds_access_raises():
    text = (helpers.GOLDEN_DIR / "kat_smoke_1d.mlir").read_text().replace(
        "bounds = #neptune_ir.bounds<lb = [1], ub = [15]>} : (!t16) -> !t16 {\n    ^bb0(%i: index, %a: !t16):\n      %m",
        "bounds = #neptune_ir.bounds<lb = [0], ub = [16]>} : (!t16) -> !t16 {\n    ^bb0(%i: index, %a: !t16):\n      %m", 1)
    m = oracle.Module.parse(text)
    with pytest.raises(oracle.OutOfBounds):
        m.call("kat_lap", np.arange(1, 17, dtype=np.float64))


def test_vectorised_evaluation_equals_the_literal_scalar_loops(kat_module):
    rng = np.random.default_rng(7)
    a, b = rng.standard_normal(16), rng.standard_normal(16)
    for sym, args in (("kat_lap", (a,)), ("kat_react", (a,)), ("kat_resid", (a, b)), ("kat_axpy", (a, b))):
        assert bits_equal(kat_module.call(sym, *args), oracle.apply_scalar_loops(kat_module, sym, *args)), sym
    m = helpers.oracle_module("3d7", (6, 5, 7))
    u = rng.standard_normal((6, 5, 7))
    assert bits_equal(m.call("lap3d", u), oracle.apply_scalar_loops(m, "lap3d", u))
    m = helpers.oracle_module("3d27", (4, 5, 6))
    u = rng.standard_normal((4, 5, 6)).astype(np.float32)
    assert bits_equal(m.call("lap27", u), oracle.apply_scalar_loops(m, "lap27", u))


@pytest.mark.parametrize("kind,shape,dtype", [("2d5", (37, 53), np.float64), ("3d7", (19, 23, 29), np.float64),
                                              ("3d27", (17, 13, 21), np.float32), ("2d5", (3, 3), np.float64),
                                              ("3d7", (3, 3, 3), np.float64)])
def test_c_restatement_equals_numpy_restatement(built_libs, kind, shape, dtype):
    u = helpers.hash_field(shape, dtype, seed=11)
    want = helpers.oracle_entry(kind, u)
    for variant in ("entry", "fused"):
        got = helpers.c_oracle_entry(kind, u, variant)
        assert bits_equal(got, want), f"{variant}\n" + mismatch_report(got, want)


def test_entry_semantics_copy_through_store_and_alias():
    u = helpers.hash_field((9, 11), np.float64, seed=3)
    m = helpers.oracle_module("2d5", u.shape)
    out = np.full_like(u, 123.0)
    res = m.call("entry", out, u)
    assert res is out                                    # unwrap of the wrapped argument: same buffer
    assert bits_equal(out[0, :], u[0, :]) and bits_equal(out[:, -1], u[:, -1])   # copy-through of input 0
    i, j = 4, 5
    want = 0.125 * ((((u[i - 1, j] + u[i + 1, j]) + u[i, j - 1]) + u[i, j + 1]) - 4.0 * u[i, j])
    assert out[i, j] == want
    fresh = m.call("lap2d", u)                           # opdef: callee-allocated result
    assert fresh is not u and bits_equal(fresh, out)


def test_store_with_bounds_and_shifted_origins():
    text = '''
#l = #neptune_ir.location<"cell">
!src = !neptune_ir.temp<element = f64, bounds = #neptune_ir.bounds<lb = [2, 0], ub = [8, 5]>, location = #l>
!dstf = !neptune_ir.field<element = f64, bounds = #neptune_ir.bounds<lb = [0, -1], ub = [10, 6]>, location = #l>
!srcf = !neptune_ir.field<element = f64, bounds = #neptune_ir.bounds<lb = [2, 0], ub = [8, 5]>, location = #l>
module {
  func.func @copy(%d: memref<?x?xf64>, %s: memref<?x?xf64>) -> memref<?x?xf64> {
    %fd = neptune_ir.wrap %d : memref<?x?xf64> -> !dstf
    %fs = neptune_ir.wrap %s : memref<?x?xf64> -> !srcf
    %t = neptune_ir.load %fs : !srcf -> !src
    neptune_ir.store %t to %fd {bounds = #neptune_ir.bounds<lb = [3, 1], ub = [7, 4]>} : !src to !dstf
    %r = neptune_ir.unwrap %fd : !dstf -> memref<?x?xf64>
    func.return %r : memref<?x?xf64>
  }
}'''
    m = oracle.Module.parse(text)
    s = np.arange(30, dtype=np.float64).reshape(6, 5)
    d = np.zeros((10, 7))
    m.call("copy", d, s)
    want = np.zeros((10, 7))
    want[3:7, 2:5] = s[1:5, 1:4]       # logical [3,7)x[1,4): src origin (2,0), dst origin (0,-1)
    assert bits_equal(d, want)


def test_hash_field_twins_agree(built_libs):
    from neptune_hip import _capi
    lib = _capi.load()
    ora = helpers.load_liboracle()
    for dtype, code, fn, ct in ((np.float64, _capi.F64, ora.ref_fill_hash_f64, C.c_double),
                                (np.float32, _capi.F32, ora.ref_fill_hash_f32, C.c_float)):
        ref = helpers.hash_field((1000,), dtype, seed=42, index_offset=5)
        buf = np.empty(1000, dtype)
        fn(buf.ctypes.data_as(C.POINTER(ct)), 1000, 5, 42)
        assert bits_equal(buf, ref)
        for i in (0, 1, 17, 999):
            assert dtype(lib.neptune_hip_hash_value(code, i + 5, 42)) == ref[i]
        assert ref.min() >= -1.0 and ref.max() < 1.0 and abs(float(ref.mean())) < 0.1

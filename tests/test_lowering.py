"""The NeptuneIR -> HIP lowering (C++ front end + emitter): runs without a GPU.
hipcc cross-compiles gfx950, so "does the emitted module build" is checked here too."""
import re
import subprocess
from pathlib import Path

import pytest

import helpers
from helpers import FIXTURE_DIR, GOLDEN_DIR
from neptune_hip import lowering

REFERENCE = Path("/root/reference/test")
NEPTUNE_OPT = helpers.REPO / "neptune-pde-solver_amd" / "bin" / "neptune-opt"


@pytest.fixture(scope="module", autouse=True)
def _built(built_libs):
    if not (lowering.LOWERING_LIB.exists() and NEPTUNE_OPT.exists()):
        subprocess.run(["make", "-C", str(helpers.REPO), "lowering"], check=True)


def _body_statements(src: str, struct: str):
    m = re.search(r"struct " + struct + r" \{.*?operator\(\)\(const A& a\) const \{(.*?)\n  \}\n\};", src, re.S)
    assert m, struct
    stmts = []
    for line in m.group(1).strip().splitlines():
        line = re.sub(r"//.*", "", line).strip()
        line = re.sub(r"\bv_", "", line)
        if line:
            stmts.append(re.sub(r"\s+", " ", line))
    return stmts


def _hexfloat_to_decimal(stmts):
    out = []
    for s in stmts:
        s = re.sub(r"-?0x[0-9a-fA-F.]+p[-+]?\d+f?", lambda m: repr(float.fromhex(m.group(0).rstrip("f"))) +
                   ("f" if m.group(0).endswith("f") else ""), s)
        out.append(s)
    return out


@pytest.mark.parametrize("fixture,tag,builtin", [("apply-2d-5pt.mlir", "lap2d_0", "Lap2D5"),
                                                 ("apply-3d-7pt.mlir", "lap3d_0", "Lap3D7"),
                                                 ("apply-3d-27pt.mlir", "lap27_0", "Lap3D27")])
def test_emitted_body_is_the_builtin_body(fixture, tag, builtin):
    """the runtime library's built-in functors (used by bench and parity tests) are exactly what the
    lowering emits for the committed fixtures: same statements, same order, same constants"""
    src, report = lowering.to_hip((FIXTURE_DIR / fixture).read_text())
    emitted = _hexfloat_to_decimal(_body_statements(src, "Body_" + tag))
    hdr = (helpers.REPO / "neptune-pde-solver_amd/csrc/runtime/builtin_bodies.hpp").read_text()
    builtin_stmts = _body_statements(hdr, builtin)
    assert emitted == builtin_stmts
    assert report["lowered"][-1] == "entry"
    assert report["applies"][0]["kernel"] == "march"
    assert report["applies"][0]["shape"] == ("box" if "27" in fixture else "star")


@pytest.mark.parametrize("kind", ["2d5", "3d7", "3d27"])
def test_fixture_generator_reproduces_the_committed_fixture(kind):
    """bench.py names a committed fixture (config.fixture) but builds its module with tools/make_stencil_mlir.py at the
    workload's size: at the fixture's nominal size the generator's text IS the committed file, comments and blank
    space aside -- so the headline's body is the fixture's body at another size, not a look-alike"""
    import make_stencil_mlir
    rank, elem, sym, fname, nominal, title = make_stencil_mlir.KINDS[kind]

    def code(text):
        lines = [re.sub(r"\s+", " ", re.sub(r"//.*", "", ln)).strip() for ln in text.splitlines()]
        return [ln for ln in lines if ln]
    assert code(make_stencil_mlir.stencil_module(kind, [nominal] * rank)) == code((FIXTURE_DIR / fname).read_text())
    # and another size changes the boxes only
    big = code(make_stencil_mlir.stencil_module(kind, [2 * nominal] * rank))
    small = code(make_stencil_mlir.stencil_module(kind, [nominal] * rank))
    differing = [(a, b) for a, b in zip(big, small) if a != b]
    assert len(big) == len(small) and differing and all("bounds<" in a for a, _ in differing)


def test_emitted_host_code_shape():
    src, report = lowering.to_hip((FIXTURE_DIR / "apply-3d-7pt.mlir").read_text())
    # exported symbols use the reference's expanded-memref ABI: 3 + 2*rank scalars per memref
    m = re.search(r'extern "C" NeptuneMemRef3D entry\((.*?)\) \{', src)
    assert m and m.group(1).count("int64_t") == 2 * (1 + 2 * 3) and m.group(1).count("void*") == 4
    # apply_linear became a call of the internal implementation, the store hands its field as destination
    assert "lap3d__impl(sc, v_u0, &v_fout, nullptr, nullptr)" in src
    assert "nl::run_store(sc, v_y, v_fout, nullptr, NEPTUNE_HIP_F64)" in src
    assert "Footprint<0, 1, 1, 1, false, true>" in src
    sig = {s["name"]: s for s in report["signatures"]}
    assert sig["entry"]["args"][0] == {"kind": "memref", "elem": "f64", "rank": 3, "shape": [-1, -1, -1], "lb": []}
    assert sig["lap3d"]["result"]["shape"] == [512, 512, 512]


def test_known_answer_module_lowers_with_scf_if_and_multiple_inputs():
    src, report = lowering.to_hip((GOLDEN_DIR / "kat_smoke_1d.mlir").read_text())
    assert report["lowered"] == ["kat_lap", "kat_react", "kat_bs", "kat_resid", "kat_axpy"]
    kinds = {a["function"]: (a["kernel"], a["shape"], a["inputs"]) for a in report["applies"]}
    # "march" = the apply is march-capable (one halo input, radius <= 2); rows narrower than a wave
    # still run on the direct kernel at launch time
    assert kinds["kat_resid"] == ("march", "star", 2) and kinds["kat_axpy"] == ("march", "pointwise", 2)
    assert "if (v_e) {" in src and "} else {" in src
    # accesses under scf.if are conditional: only the unconditional ones enter the plan-time bounds check
    # (reach = {most negative offsets}, {most positive}; hi < lo: the input is not accessed unconditionally)
    assert "neptune_hip::Reach kTopRadius_kat_resid_0 = {{{1, 1, 1}, {1, 1, 1}, {1, 1, 1}, {1, 1, 1}}, {{-1, -1, -1}, {-1, -1, -1}" in src
    # ... and an unconditional one-sided access keeps its sign: a face field read at [0] and [+1] reaches (0, +1)
    import test_ownbox_gpu as ob
    src2, _ = lowering.to_hip(ob.case_text("staggered_1d"))
    assert "neptune_hip::Reach kTopRadius_resid_0 = {{{0, 1, 1}, {0, 1, 1}, {-2, 1, 1}, {1, 1, 1}}, {{0, -1, -1}, {1, -1, -1}, {2, -1, -1}, {-1, -1, -1}}}" in src2
    assert "a.template idx<0>()" in src


@pytest.mark.skipif(not REFERENCE.exists(), reason="reference tree not mounted (GPU box)")
def test_reference_inputs_lower_or_fail_like_the_reference():
    for name, lowered in (("smoke_time_advance.mlir", ["ac_lap", "ac_A"]), ("smoke_time_advance_bs.mlir", ["bs_A"]),
                          ("smoke_time_advance_nonlinear.mlir", ["ac_residual"])):
        _, report = lowering.to_hip((REFERENCE / "smoke_tests" / name).read_text())
        # @entry holds neptune_ir.time_advance: solver surface, stays on the host path ...
        assert report["skipped"][0]["symbol"] == "entry" and "time_advance" in report["skipped"][0]["reason"]
        # ... but the stencil value it hands to the solver op (smoke_time_advance.mlir:59-70: %ustar) is outlined
        outlined = [o["symbol"] for o in report["outlined"]]
        assert report["lowered"] == lowered + outlined
        if name == "smoke_time_advance.mlir":
            assert report["outlined"] == [{"symbol": "entry__stencil_0", "function": "entry", "value": "%ustar", "line": 59}]
    # old-style regions (^bb0(%i0: index) capturing the outer temp) fail ApplyOp::verify in the
    # reference too (NeptuneIRVerifier.cpp:150-168); same diagnostic text
    for name in ("smoke.mlir", "smoke_apply.mlir", "smoke_assemble_matrix.mlir"):
        with pytest.raises(lowering.LoweringError, match=r"block arg count must be \(bounds rank \+ number of inputs\) = 2, but got 1"):
            lowering.to_hip((REFERENCE / "smoke_tests" / name).read_text())


def test_stencil_part_of_a_function_with_a_solver_op_is_outlined():
    """@entry of tests/golden/kat_outline_1d.mlir (the shape of smoke_time_advance.mlir:53-84) is not lowered -- it holds an
    implicit time_advance -- but %ustar, the apply result the solver op consumes, becomes the exported symbol
    entry__stencil_0 with @entry's own memref arguments and a memref result"""
    text = (GOLDEN_DIR / "kat_outline_1d.mlir").read_text()
    src, report = lowering.to_hip(text)
    assert report["lowered"] == ["ol_lap", "ol_A", "entry__stencil_0"]
    assert report["skipped"][0]["symbol"] == "entry"
    assert report["outlined"] == [{"symbol": "entry__stencil_0", "function": "entry", "value": "%ustar", "line": 47}]
    sig = {x["name"]: x for x in report["signatures"]}["entry__stencil_0"]
    assert [a["kind"] for a in sig["args"]] == ["memref", "memref"] and sig["result"]["kind"] == "temp" and sig["result"]["shape"] == [16]
    assert re.search(r'extern "C" NeptuneMemRef1D entry__stencil_0\(', src)
    assert "time_advance" not in src.split("// ---- @entry__stencil_0")[1].split("extern")[0]
    # two values consumed by solver ops -> two symbols; a value computed from a solver result is not outlined;
    # a store ahead of the producer (it could feed the producer through memory) blocks the outlining
    two = text.replace("    %dt = arith.constant 1.0e-2 : f64\n    %u1 =", """    %w = neptune_ir.apply_linear @ol_lap(%u0) : (!t) -> !t
    %m = neptune_ir.assemble_matrix @ol_A : !neptune_ir.matrix
    %sol = neptune_ir.solve_linear %m, %w {solver = "cg", tol = 1.0e-8} : !neptune_ir.matrix, !t -> !t
    %late = neptune_ir.apply_linear @ol_lap(%sol) : (!t) -> !t
    %dt = arith.constant 1.0e-2 : f64
    %u1 =""")
    assert two != text
    _, rep2 = lowering.to_hip(two)
    assert [(o["symbol"], o["value"]) for o in rep2["outlined"]] == [("entry__stencil_0", "%w"), ("entry__stencil_1", "%ustar")]
    blocked = text.replace("    %u0 = neptune_ir.load %fin : !f -> !t\n", "    %u0 = neptune_ir.load %fin : !f -> !t\n    neptune_ir.store %u0 to %fout : !t to !f\n")
    assert blocked != text
    _, rep3 = lowering.to_hip(blocked)
    assert rep3["outlined"] == [] and rep3["lowered"] == ["ol_lap", "ol_A"]


BAD = {
    "0-D apply": ("bounds = #neptune_ir.bounds<lb = [1], ub = [15]>} : (!t16) -> !t16 {\n    ^bb0(%i: index, %a: !t16):\n      %m",
                  "bounds = #neptune_ir.bounds<lb = [], ub = []>} : (!t16) -> !t16 {\n    ^bb0(%i: index, %a: !t16):\n      %m",
                  "0-D apply not supported"),
    "index arg": ("^bb0(%i: index, %a: !t16):\n      %m", "^bb0(%i: f64, %a: !t16):\n      %m", "region arg #0 must be index"),
    "yield type": ("neptune_ir.yield %v3 : f64", "neptune_ir.yield %i : index", "yield operand type must equal apply result element type"),
    "offset rank": ("neptune_ir.access %a[-1] : !t16 -> f64\n      %z = neptune_ir.access %a[0] : !t16 -> f64\n      %p = neptune_ir.access %a[1] : !t16 -> f64\n      %k2",
                    "neptune_ir.access %a[-1, 0] : !t16 -> f64\n      %z = neptune_ir.access %a[0] : !t16 -> f64\n      %p = neptune_ir.access %a[1] : !t16 -> f64\n      %k2",
                    "offsets rank must match apply bounds rank"),
    "linear mul": ("%v0 = arith.mulf %k2, %z : f64", "%v0 = arith.mulf %z, %z : f64", "MulFOp in linear region must multiply by a constant"),
    "linear op": ("%v1 = arith.subf %m, %v0 : f64", "%v1 = arith.divf %m, %v0 : f64", "op not allowed inside apply for linear_opdef"),
    "undefined": ("%v2 = arith.addf %v1, %p : f64", "%v2 = arith.addf %v1, %nope : f64", "use of undefined value %nope"),
}


@pytest.mark.parametrize("case", sorted(BAD))
def test_verifier_diagnostics(case):
    old, new, msg = BAD[case]
    text = (GOLDEN_DIR / "kat_smoke_1d.mlir").read_text()
    assert old in text
    with pytest.raises(lowering.LoweringError, match=re.escape(msg)):
        lowering.verify(text.replace(old, new, 1))
    lowering.verify(text)


def test_neptune_opt_cli(tmp_path):
    out = tmp_path / "m.hip"
    p = subprocess.run([str(NEPTUNE_OPT), str(FIXTURE_DIR / "apply-2d-5pt.mlir"), "--neptuneir-to-hip", "-o", str(out), "--report"],
                       capture_output=True, text=True)
    assert p.returncode == 0 and '"lowered": ["lap2d", "entry"]' in p.stdout
    assert "struct Body_lap2d_0" in out.read_text()
    # the reference's own invocation is recognised and redirected, not silently accepted
    p = subprocess.run([str(NEPTUNE_OPT), str(FIXTURE_DIR / "apply-2d-5pt.mlir"), "--neptuneir-to-llvm"], capture_output=True, text=True)
    assert p.returncode == 2 and "--neptuneir-to-hip" in p.stderr
    bad = tmp_path / "bad.mlir"
    bad.write_text("module { func.func @f() { neptune_ir.bogus } }")
    p = subprocess.run([str(NEPTUNE_OPT), str(bad), "--verify-only"], capture_output=True, text=True)
    assert p.returncode == 1 and "error:" in p.stderr


def test_emitted_modules_compile_for_gfx950(tmp_path, monkeypatch):
    """hipcc cross-compiles without a GPU: the emitted translation units build and export the
    symbols the report names (loading them needs the HIP runtime but no device)"""
    monkeypatch.setenv("NEPTUNE_CACHE_DIR", str(tmp_path))
    for text in ((GOLDEN_DIR / "kat_smoke_1d.mlir").read_text(), helpers.stencil_module("3d7", (16, 16, 128))):
        mod = lowering.compile_module(text)
        assert mod.path.parent == tmp_path and mod.path.name == f"neptune_kernel_{lowering.module_hash(text)}.so"
        for sym in mod.report["lowered"]:
            assert hasattr(mod.lib, sym)
        again = lowering.compile_module(text)     # second call is a cache hit (same file, no recompile)
        assert again.path == mod.path
    # the heaviest footprint the march kernel accepts in 2-D (four halo inputs of radius 2) must still fit the
    # 160 KiB of LDS of a CU -- hipcc rejects a kernel that does not
    import test_multihalo_gpu as mh
    shape, elem, nin, acc, margin, _ = mh.CASES["four_radius2_2d_f32"]
    heavy = lowering.compile_module(mh.module_text(shape, elem, nin, acc, [margin] * 2, [n - margin for n in shape]))
    assert hasattr(heavy.lib, "resid") and heavy.report["applies"][0]["kernel"] == "march"


NORM = '''
#l = #neptune_ir.location<"cell">
!t = !neptune_ir.temp<element = f64, bounds = #neptune_ir.bounds<lb = [0, 0], ub = [{n0}, {n1}]>, location = #l>
!f = !neptune_ir.field<element = f64, bounds = #neptune_ir.bounds<lb = [0, 0], ub = [{n0}, {n1}]>, location = #l>
module {{
  func.func @norm2(%a: memref<?x?xf64>) -> f64 {{
    %fa = neptune_ir.wrap %a : memref<?x?xf64> -> !f
    %u = neptune_ir.load %fa : !f -> !t
    %sq = neptune_ir.apply(%u) attributes {{bounds = #neptune_ir.bounds<lb = [0, 0], ub = [{n0}, {n1}]>}} : (!t) -> !t {{
      ^bb0(%i: index, %j: index, %x: !t):
        %v = neptune_ir.access %x[0, 0] : !t -> f64
        %p = arith.mulf %v, %v : f64
        neptune_ir.yield %p : f64
    }}
    %s = neptune_ir.reduce %sq in #neptune_ir.bounds<lb = [1, 1], ub = [{m0}, {m1}]> {{kind = "sum"}} : !t -> f64
    %r = math.sqrt %s : f64
    func.return %r : f64
  }}
}}
'''


def test_reduce_and_scalar_results_lower():
    """neptune_ir.reduce {kind = "sum"} (DataflowLowering.cpp:589-698) + scalar
    arithmetic and a scalar function result"""
    text = NORM.format(n0=6, n1=8, m0=5, m1=7)
    src, report = lowering.to_hip(text)
    assert 'extern "C" double norm2(' in src
    # the single-use apply result is evaluated inside the reduction kernel ...
    assert "nl::run_apply_reduce_sum<Body_norm2_0, double, 2, 1, FP_norm2_0>(sc, Body_norm2_0{}" in src
    assert "neptune_hip::ops::sqrt(v_s)" in src and report["applies"][0]["kernel"] == "reduce"
    # ... a reduce of anything else (here: the loaded field itself) reads its operand from memory
    src2, _ = lowering.to_hip(text.replace("neptune_ir.reduce %sq in", "neptune_ir.reduce %u in"))
    assert "nl::run_reduce_sum(sc, v_u, &kBox" in src2 and "run_apply_reduce_sum" not in src2
    assert report["signatures"][0]["result"]["kind"] == "scalar"
    with pytest.raises(lowering.LoweringError, match='MVP reduce only supports kind="sum"'):
        lowering.verify(text.replace('kind = "sum"', 'kind = "max"'))
    # oracle: serial left-to-right sum in row-major order, exactly
    import math
    import numpy as np
    from helpers import oracle
    a = helpers.hash_field((6, 8), np.float64, seed=2)
    acc = 0.0
    for v in (a[1:5, 1:7] * a[1:5, 1:7]).ravel():
        acc = acc + v
    assert oracle.Module.parse(text).call("norm2", a) == math.sqrt(acc)


def test_explicit_time_advance_lowers_and_implicit_stays_on_the_host():
    text = (helpers.REPO / "tests/mlir_tests/time_stepping/explicit-heat-2d.mlir").read_text()
    src, report = lowering.to_hip(text)
    assert report["lowered"] == ["lap", "step"]
    # the rhs opdef is one apply of the state: rhs and axpy run as ONE kernel over the state
    assert "lap__impl(sc, v_u0" not in src
    assert "nl::run_apply<neptune_hip::ops::EulerFused<Body_lap_0, double, 2>, double, 2, 1, FP_lap_0>" in src
    assert "neptune_hip::Reach kTopRadius_step_ta0 = {{{-1, -1, 1}, {1, 1, 1}" in src and "}}, {{1, 1, -1}, {-1, -1, -1}" in src
    assert [a["inputs"] for a in report["applies"] if a["function"] == "step"] == [1]
    # a rhs opdef made of two applies is not fusable: call @rhs, then the axpy apply
    src2, rep2s = lowering.to_hip((helpers.REPO / "tests/mlir_tests/time_stepping/explicit-twostage-3d.mlir").read_text())
    assert "rhs__impl(sc, v_u0, nullptr, nullptr, nullptr)" in src2 and "EulerAxpy<double, 3>{(double)v_dt}" in src2
    assert [a["inputs"] for a in rep2s["applies"] if a["function"] == "step"] == [2]
    # oracle: out = s + dt*k, two roundings, whole box (copy-through cells see k = s)
    import numpy as np
    from helpers import oracle
    u = helpers.hash_field((12, 128), np.float64, seed=3)
    out = np.zeros_like(u)
    oracle.Module.parse(text).call("step", out, u)
    i, j = 4, 9
    lap = (((u[i - 1, j] + u[i + 1, j]) + u[i, j - 1]) + u[i, j + 1]) - 4.0 * u[i, j]
    assert out[i, j] == u[i, j] + 0.1 * lap and out[0, 5] == u[0, 5] + 0.1 * u[0, 5]
    # implicit methods need the solver runtime: the function is reported, not lowered
    _, rep2 = lowering.to_hip(text.replace("method = 0 : i32, rhs = @lap", 'method = 2 : i32, system = @lap, solver = "gmres"'))
    assert rep2["lowered"] == ["lap"] and rep2["skipped"][0]["symbol"] == "step"
    with pytest.raises(lowering.LoweringError, match="rhs must reference linear_opdef or nonlinear_opdef"):
        lowering.verify(text.replace("rhs = @lap", "rhs = @nope"))


def test_several_halo_inputs_get_a_mask_and_shared_radii():
    """inputs 1 and 2 read at offsets, input 0 only at the centre: Footprint<first halo input, shared radii..., mask>;
    radius 2 with two halo inputs in 3-D is left to the direct kernel"""
    import test_multihalo_gpu as mh
    shape, elem, nin, accesses, _, _ = mh.CASES["point0_then_two_stars"]
    src, rep = lowering.to_hip(mh.module_text(shape, elem, nin, accesses, [1, 1, 1], [n - 1 for n in shape]))
    assert "neptune_hip::Footprint<1, 1, 1, 1, false, true, 0x6u>" in src
    assert rep["applies"][0]["kernel"] == "march"
    shape, elem, nin, accesses, _, _ = mh.CASES["radius2_pair_3d"]
    src, rep = lowering.to_hip(mh.module_text(shape, elem, nin, accesses, [2, 2, 2], [n - 2 for n in shape]))
    assert "neptune_hip::Footprint<0, 2, 2, 2, false, true, 0x3u>" in src      # two rings + two LDS windows (plane kernel)
    assert rep["applies"][0]["kernel"] == "march"
    shape, elem, nin, accesses, _, _ = mh.CASES["radius5_pair_3d"]
    src, rep = lowering.to_hip(mh.module_text(shape, elem, nin, accesses, [5, 5, 5], [n - 5 for n in shape]))
    assert "neptune_hip::Footprint<-1, 0, 0, 0, false, false>" in src
    assert rep["applies"][0]["kernel"] == "direct"
    shape, elem, nin, accesses, _, _ = mh.CASES["four_halo_inputs_2d"]
    src, rep = lowering.to_hip(mh.module_text(shape, elem, nin, accesses, [1, 1], [n - 1 for n in shape]))
    assert "neptune_hip::Footprint<0, 1, 0, 1, false, true, 0xfu>" in src


def test_footprint_limits_of_the_march_kernel():
    """which footprints the emitter hands to the march kernel (emit_hip.cpp analyze_apply): stars up to radius 4 (3-D stars
    of one halo input up to radius 8: plane-in-LDS kernel), in 1-D / 2-D a K
    radius of at most two 16-byte lane vectors, 3-D boxes of radius 1, 2-D boxes of radius 2, one wide halo input;
    every test_multihalo_gpu case is listed with the kernel it expects"""
    import test_multihalo_gpu as mh
    for name, (shape, elem, nin, accesses, margin, kernel) in mh.CASES.items():
        rank = len(shape)
        _, rep = lowering.to_hip(mh.module_text(shape, elem, nin, accesses, [margin] * rank, [n - margin for n in shape]))
        assert rep["applies"][0]["kernel"] == kernel, name

    def kernel_of(shape, elem, accesses, margin):
        rank = len(shape)
        src, rep = lowering.to_hip(mh.module_text(shape, elem, 1, accesses, [margin] * rank, [n - margin for n in shape]))
        return rep["applies"][0]["kernel"], src

    k, src = kernel_of((20, 20, 128), "f32", [(0, o) for o in mh.star(3, 4)], 4)
    assert k == "march" and "neptune_hip::Footprint<0, 4, 4, 4, false, true>" in src
    # 3-D stars of one halo input reach radius 8 (the plane-in-LDS kernel holds only the ring of own cells in registers)
    k, src = kernel_of((24, 24, 128), "f32", [(0, o) for o in mh.star(3, 5) if max(map(abs, o)) in (0, 5)], 5)
    assert k == "march" and "neptune_hip::Footprint<0, 5, 5, 5, false, true>" in src
    assert kernel_of((40, 40, 128), "f64", [(0, o) for o in mh.star(3, 8)], 8)[0] == "march"
    assert kernel_of((40, 40, 128), "f64", [(0, o) for o in mh.star(3, 9) if max(map(abs, o)) in (0, 9)], 9)[0] == "direct"
    two = [(0, o) for o in mh.star(3, 5)] + [(1, o) for o in mh.star(3, 1)[1:]]
    src, rep = lowering.to_hip(mh.module_text((24, 24, 128), "f64", 2, two, [5] * 3, [19] * 3))
    assert rep["applies"][0]["kernel"] == "direct"                  # two halo inputs: up to radius 4
    # 1-D: the K radius may reach two lane vectors (8 f32 cells, 4 f64 cells)
    assert kernel_of((1024,), "f32", [(0, o) for o in mh.star(1, 8)], 8)[0] == "march"
    assert kernel_of((1024,), "f32", [(0, o) for o in mh.star(1, 9)], 9)[0] == "direct"
    assert kernel_of((1024,), "f64", [(0, o) for o in mh.star(1, 4)], 4)[0] == "march"
    assert kernel_of((1024,), "f64", [(0, o) for o in mh.star(1, 5)], 5)[0] == "direct"
    # boxes: 27 points in 3-D, 5x5 in 2-D
    box3 = [(0, (a, b, c)) for a in (-2, 0, 2) for b in (-1, 1) for c in (-2, 2)]
    assert kernel_of((16, 16, 128), "f64", box3, 2)[0] == "march"       # radius-2 boxes: every live plane in LDS
    box3 = [(0, (a, b, c)) for a in (-3, 0, 3) for b in (-1, 1) for c in (-2, 2)]
    assert kernel_of((16, 16, 128), "f64", box3, 3)[0] == "direct"
    box2 = [(0, (a, b)) for a in range(-3, 4) for b in (-3, 3)]
    assert kernel_of((32, 256), "f64", box2, 3)[0] == "march"          # 2-D beyond radius 2 / 4: the LDS tile kernel, up to radius 8
    assert kernel_of((40, 256), "f64", [(0, o) for o in mh.star(2, 9) if max(map(abs, o)) in (0, 9)], 9)[0] == "direct"


def test_every_apply_gets_a_geometry_level_entry():
    """the module's counterpart of neptune_hip_apply_builtin: explicit geometry, region, stream and launch cfg"""
    src, report = lowering.to_hip((FIXTURE_DIR / "apply-3d-13pt.mlir").read_text())
    a = report["applies"][0]
    assert a["geom_symbol"] == "lap13_0__geom" and a["halo0"] == 2 and a["elem"] == "f64"
    assert 'extern "C" int lap13_0__geom(const neptune_hip_apply_geom_t* g, const void* const* in, void* out, void* stream,' in src
    assert "neptune_hip::launch_apply<Body_lap13_0, double, 3, 1, FP_lap13_0>(Body_lap13_0{}, g, in, out, (hipStream_t)stream, cfg)" in src
    # applies folded into a reduce have no kernel of their own, hence no entry
    _, rep2 = lowering.to_hip(NORM.format(n0=6, n1=8, m0=5, m1=7))
    assert rep2["applies"][0]["geom_symbol"] == ""


def test_front_end_survives_mutated_inputs():
    """truncated / spliced / flipped / number-swapped versions of every fixture and of random generator modules: neptune-opt may
    accept or reject them but must terminate with exit code 0 or 1 (tools/fuzz_frontend.py; its sanitizer build found a
    parser loop that never ended on a truncated attribute dictionary and uncaught range errors on ops cut short)"""
    import sys
    sys.path.insert(0, str(helpers.REPO / "tools"))
    import fuzz_frontend
    originals, mutants, codes, findings = fuzz_frontend.run(NEPTUNE_OPT, mutants_per_input=4, seed=2, timeout=20.0)
    assert originals >= 40 and mutants == 4 * originals
    assert not findings, findings[0][:4]
    assert set(codes) <= {0, 1}
    truncated = "module {\n  func.func @f(%a: memref<?xf64>) -> memref<?xf64> attributes {"
    with pytest.raises(lowering.LoweringError, match="unexpected end of input"):
        lowering.verify(truncated)


ELEMENTARY = '''
#l = #neptune_ir.location<"cell">
!t = !neptune_ir.temp<element = {elem}, bounds = #neptune_ir.bounds<lb = [0, 0], ub = [{n0}, {n1}]>, location = #l>
module {{
  // Arrhenius-type reaction term on top of a 5-point diffusion: elementary functions in an apply body
  neptune_ir.nonlinear_opdef @react : (!t) -> !t {{
  ^bb0(%u: !t):
    %r = neptune_ir.apply(%u) attributes {{bounds = #neptune_ir.bounds<lb = [1, 1], ub = [{m0}, {m1}]>}} : (!t) -> !t {{
      ^bb0(%i: index, %j: index, %a: !t):
        %c = neptune_ir.access %a[0, 0] : !t -> {elem}
        %n = neptune_ir.access %a[-1, 0] : !t -> {elem}
        %s = neptune_ir.access %a[1, 0] : !t -> {elem}
        %w = neptune_ir.access %a[0, -1] : !t -> {elem}
        %e = neptune_ir.access %a[0, 1] : !t -> {elem}
        %half = arith.constant 0.5 : {elem}
        %two = arith.constant 2.0 : {elem}
        %ns = arith.addf %n, %s : {elem}
        %we = arith.addf %w, %e : {elem}
        %nb = arith.addf %ns, %we : {elem}
        %ab = math.absf %c : {elem}
        %den = arith.addf %ab, %half : {elem}
        %inv = arith.divf %half, %den : {elem}
        %neg = arith.negf %inv : {elem}
        %arr = math.exp %neg : {elem}
        %sn = math.sin %nb : {elem}
        %cs = math.cos %c : {elem}
        %th = math.tanh %nb : {elem}
        %lg = math.log %den : {elem}
        %pw = math.powf %den, %two : {elem}
        %t0 = arith.mulf %arr, %sn : {elem}
        %t1 = arith.addf %t0, %cs : {elem}
        %t2 = arith.mulf %th, %lg : {elem}
        %t3 = arith.addf %t1, %t2 : {elem}
        %t4 = arith.addf %t3, %pw : {elem}
        neptune_ir.yield %t4 : {elem}
    }}
    neptune_ir.return %r : !t
  }}
}}
'''


def test_elementary_functions_lower_and_are_flagged_inexact():
    """math.exp / log / sin / cos / tanh / powf: not exactly specified (libm in the reference's lowering, the device math library
    here), so such bodies are lowered but reported as "exact": false; everything else stays "exact": true"""
    text = ELEMENTARY.format(elem="f64", n0=8, n1=128, m0=7, m1=127)
    lowering.verify(text)
    src, report = lowering.to_hip(text)
    for fn in ("exp", "log", "sin", "cos", "tanh"):
        assert f"neptune_hip::ops::{fn}(" in src
    assert "neptune_hip::ops::powf(v_den, v_two)" in src
    assert report["applies"][0]["exact"] is False and report["applies"][0]["kernel"] == "march"
    _, rep7 = lowering.to_hip((FIXTURE_DIR / "apply-3d-7pt.mlir").read_text())
    assert rep7["applies"][0]["exact"] is True
    import numpy as np
    from helpers import oracle
    u = helpers.hash_field((8, 128), np.float64, seed=1)
    got = oracle.Module.parse(text).call("react", u)
    i, j = 3, 17
    nb = (u[i - 1, j] + u[i + 1, j]) + (u[i, j - 1] + u[i, j + 1])
    den = abs(u[i, j]) + 0.5
    want = ((np.exp(-(0.5 / den)) * np.sin(nb) + np.cos(u[i, j])) + np.tanh(nb) * np.log(den)) + np.power(den, 2.0)
    assert got[i, j] == want and got[0, 5] == u[0, 5]
    with pytest.raises(lowering.LoweringError, match="unsupported operation 'math.erf'"):
        lowering.verify(text.replace("math.tanh %nb", "math.erf %nb"))


def test_scalar_results_say_what_they_mean_on_a_slab():
    """the lowering reports, per scalar-returning function, whether the value is a bare reduce (the ranks' partial sums
    add up), the same on every rank, or computed FROM a reduce (only right on one rank) -- what ShardedModule needs to
    combine or refuse it; the derived kind also aborts inside the lowered function when ghost planes are present"""
    text = NORM.format(n0=6, n1=8, m0=5, m1=7)
    src, report = lowering.to_hip(text)
    assert report["signatures"][0]["result"]["scalar"] == "derived"          # sqrt(reduce)
    assert "slab mode: the returned scalar is computed from a reduce result" in src
    bare = text.replace("%r = math.sqrt %s : f64\n", "").replace("func.return %r", "func.return %s")
    src_b, rep_b = lowering.to_hip(bare)
    assert rep_b["signatures"][0]["result"]["scalar"] == "partial_sum" and "slab mode: the returned scalar" not in src_b
    const = text.replace("%r = math.sqrt %s : f64", "%r = arith.constant 2.5 : f64")
    _, rep_c = lowering.to_hip(const)
    assert rep_c["signatures"][0]["result"]["scalar"] == "uniform"
    # reduce + constant is derived too (the constant would be added once per rank)
    plus = text.replace("%r = math.sqrt %s : f64", "%k = arith.constant 1.0 : f64\n    %r = arith.addf %s, %k : f64")
    _, rep_p = lowering.to_hip(plus)
    assert rep_p["signatures"][0]["result"]["scalar"] == "derived"


def test_sharded_module_refuses_scalars_derived_from_a_partial_sum():
    """sqrt(reduce(...)) summed over ranks is silently wrong; the call is refused before anything runs"""
    from neptune_hip import slab

    class FakeModule:
        signatures = {"norm2": {"name": "norm2", "args": [], "result": {"kind": "scalar", "elem": "f64", "rank": 0, "scalar": "derived"}}}

        def call(self, *a):
            raise AssertionError("must not be called")
    sl = slab.decompose(([0, 0], [8, 8]), 1, 0, 2)
    sm = slab.ShardedModule.__new__(slab.ShardedModule)
    sm.module, sm.slab, sm.group, sm._lib = FakeModule(), sl, None, None
    with pytest.raises(ValueError, match="partial sum"):
        sm.call("norm2")


CALLEE_READS_LATER = '''
#l = #neptune_ir.location<"cell">
!t = !neptune_ir.temp<element = f64, bounds = #neptune_ir.bounds<lb = [0], ub = [256]>, location = #l>
!f = !neptune_ir.field<element = f64, bounds = #neptune_ir.bounds<lb = [0], ub = [256]>, location = #l>
module {
  neptune_ir.nonlinear_opdef @g : (!t, !t) -> !t {
  ^bb0(%x: !t, %y: !t):
    %a = neptune_ir.apply(%x) attributes {bounds = #neptune_ir.bounds<lb = [1], ub = [255]>} : (!t) -> !t {
      ^bb0(%i: index, %xi: !t):
        %l = neptune_ir.access %xi[-1] : !t -> f64
        %r = neptune_ir.access %xi[1] : !t -> f64
        %s = arith.addf %l, %r : f64
        neptune_ir.yield %s : f64
    }
    %b = neptune_ir.apply(%y) attributes {bounds = #neptune_ir.bounds<lb = [1], ub = [255]>} : (!t) -> !t {
      ^bb0(%i: index, %yi: !t):
        %l = neptune_ir.access %yi[-1] : !t -> f64
        neptune_ir.yield %l : f64
    }
    PLACEHOLDER
  }
}
'''


def test_destination_is_only_forwarded_to_the_last_producer_of_a_callee():
    """ADVICE r1 (store elision through calls): the caller's destination field may be written directly only by a
    producer that is the last thing the callee computes, and never when it overlaps one of the callee's arguments"""
    # returned producer %a is followed by another apply (%b reads %y, which may alias the caller's destination)
    early = CALLEE_READS_LATER.replace("PLACEHOLDER", "neptune_ir.return %a : !t")
    src, _ = lowering.to_hip(early)
    impl = src[src.index("static nl::Val g__impl(nl::Scope& sc"):]
    impl = impl[:impl.index("\n}\n")]
    assert "kTopRadius_g_0, nullptr," in impl and ", dest," not in impl          # %a gets a private result
    # returned producer %b IS the last op: it may take the destination, guarded by the overlap check on the arguments
    last = CALLEE_READS_LATER.replace("PLACEHOLDER", "neptune_ir.return %b : !t")
    src2, _ = lowering.to_hip(last)
    impl2 = src2[src2.index("static nl::Val g__impl(nl::Scope& sc"):]
    impl2 = impl2[:impl2.index("\n}\n")]
    assert "if (dest && (nl::overlaps(*dest, v_x) || nl::overlaps(*dest, v_y))) dest = nullptr;" in impl2
    assert "kTopRadius_g_1, dest," in impl2 and "kTopRadius_g_0, nullptr," in impl2


TENSOR_CASTS = '''
#l = #neptune_ir.location<"cell">
!t = !neptune_ir.temp<element = f64, bounds = #neptune_ir.bounds<lb = [2, 0], ub = [10, 128]>, location = #l>
module {
  neptune_ir.linear_opdef @A : (!t) -> !t {
  ^bb0(%u: !t):
    %r = neptune_ir.apply(%u) attributes {bounds = #neptune_ir.bounds<lb = [3, 1], ub = [9, 127]>} : (!t) -> !t {
      ^bb0(%i: index, %j: index, %a: !t):
        %n = neptune_ir.access %a[-1, 0] : !t -> f64
        %s = neptune_ir.access %a[1, 0] : !t -> f64
        %v = arith.addf %n, %s : f64
        neptune_ir.yield %v : f64
    }
    neptune_ir.return %r : !t
  }
  func.func @f(%x: tensor<8x128xf64>) -> tensor<8x128xf64> {
    %a = neptune_ir.from_tensor %x : tensor<8x128xf64> -> !t
    %b = neptune_ir.apply_linear @A(%a) : (!t) -> !t
    %c = neptune_ir.as_tensor %b : !t -> tensor<8x128xf64>
    func.return %c : tensor<8x128xf64>
  }
}
'''


def test_as_tensor_and_from_tensor_are_casts_of_the_same_buffer():
    """neptune_ir.as_tensor / from_tensor (NeptuneIROps.td:540-591) stay casts in the reference's dataflow lowering
    (DataflowLowering.cpp:705-733); here a ranked tensor is read like a static memref and both ops alias the buffer"""
    import numpy as np
    src, report = lowering.to_hip(TENSOR_CASTS)
    assert 'sc.alias(v_x, kBox' in src and '"neptune_ir.from_tensor"' in src and '"neptune_ir.as_tensor"' in src
    sig = {s["name"]: s for s in report["signatures"]}["f"]
    assert sig["args"][0]["kind"] == "memref" and sig["args"][0]["shape"] == [8, 128]
    with pytest.raises(lowering.LoweringError, match="tensor shape must equal the extents"):
        lowering.verify(TENSOR_CASTS.replace("%c = neptune_ir.as_tensor %b : !t -> tensor<8x128xf64>",
                                             "%c = neptune_ir.as_tensor %b : !t -> tensor<8x64xf64>")
                        .replace("func.return %c : tensor<8x128xf64>", "func.return %c : tensor<8x64xf64>")
                        .replace("-> tensor<8x128xf64> {", "-> tensor<8x64xf64> {"))
    with pytest.raises(lowering.LoweringError, match="element type mismatch"):
        lowering.verify(TENSOR_CASTS.replace("%a = neptune_ir.from_tensor %x : tensor<8x128xf64> -> !t",
                                             "%a = neptune_ir.from_tensor %x : tensor<8x128xf32> -> !t")
                        .replace("@f(%x: tensor<8x128xf64>)", "@f(%x: tensor<8x128xf32>)"))
    # the oracle agrees on what the function computes: the operator applied to the buffer read with origin [2, 0]
    from helpers import oracle
    m = oracle.Module.parse(TENSOR_CASTS)
    x = helpers.hash_field((8, 128), np.float64, seed=9)
    got = m.call("f", x)
    want = x.copy()
    want[1:7, 1:127] = x[0:6, 1:127] + x[2:8, 1:127]
    assert helpers.bits_equal(np.asarray(got), want)


def test_every_committed_module_cross_compiles_for_gfx950(tmp_path, monkeypatch):
    """every .mlir this repository commits (conversion fixtures, time-stepping and nonlinear fixtures, the golden KAT modules)
    goes through lower + hipcc --offload-arch=gfx950 here, without a GPU: a change of the runtime headers that breaks ANY
    emitted construct (a fused or unfused time step, a reduce, a rank-4 apply, an outlined function) shows up in the CPU suite
    instead of at the first GPU run"""
    import glob
    monkeypatch.setenv("NEPTUNE_CACHE_DIR", str(tmp_path))
    paths = sorted(glob.glob(str(helpers.REPO / "tests/mlir_tests/**/*.mlir"), recursive=True)) + sorted(glob.glob(str(GOLDEN_DIR / "*.mlir")))
    assert len(paths) >= 10
    texts = [Path(f).read_text() for f in paths]
    helpers.prefetch_modules(texts, workers=6)
    built = {lowering.module_hash(t) for t in texts}
    missing = [f for f, t in zip(paths, texts) if not (tmp_path / f"neptune_kernel_{lowering.module_hash(t)}.so").exists()]
    logs = "".join(p.read_text()[:1500] for p in tmp_path.glob("*.log") if "error" in p.read_text())
    assert not missing, f"did not compile: {missing}\n{logs}"
    assert len(built) == len(set(texts))


def test_absurd_bounds_and_offsets_are_refused_before_any_arithmetic_overflows():
    """found by the sanitizer fuzz of round 3 (tools/fuzz_frontend.py): a bounds coordinate of 2^63 made `ub - lb` overflow in
    the emitter; coordinates beyond 2^40, boxes beyond 2^62 cells and access offsets beyond 2^20 are diagnosed by the front end"""
    from neptune_hip import lowering
    text = (helpers.FIXTURE_DIR / "apply-2d-5pt.mlir").read_text()
    lowering.verify(text)
    with pytest.raises(lowering.LoweringError, match=r"bounds coordinate outside the supported range \[-2\^40, 2\^40\]"):
        lowering.verify(text.replace("ub = [1024, 1024]", "ub = [9223372036854775808, 1024]"))
    with pytest.raises(lowering.LoweringError, match=r"bounds coordinate outside the supported range"):
        lowering.verify(text.replace("lb = [0, 0]", "lb = [-99999999999999, 0]", 1))
    with pytest.raises(lowering.LoweringError, match="bounds describe more than 2\\^62 cells"):
        lowering.verify(text.replace("ub = [1024, 1024]", "ub = [1099511627776, 1099511627776]"))
    with pytest.raises(lowering.LoweringError, match=r"offset outside the supported range \[-2\^20, 2\^20\]"):
        lowering.verify(text.replace("[-1, 0]", "[-99999999999, 0]"))

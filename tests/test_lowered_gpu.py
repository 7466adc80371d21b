"""Lowered modules on the GPU: NeptuneIR text -> neptune-opt lowering -> hipcc -> ctypes, called
through the reference's expanded-memref ABI, bit-exact against the oracle."""
import subprocess
import sys

import numpy as np
import pytest

import helpers
from helpers import bits_equal, mismatch_report, oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(built_libs, tmp_path_factory):
    import os
    import torch
    assert torch.cuda.is_available()
    os.environ["NEPTUNE_CACHE_DIR"] = str(tmp_path_factory.mktemp("neptune_cache"))
    from neptune_hip import lowering
    return lowering, torch


def _vec(h):
    return np.array([float.fromhex(x) for x in h])


def test_reference_smoke_bodies_known_answers(env):
    """KAT-1..5 through the full lowering: includes scf.if with index compares and 2-input applies"""
    lowering, torch = env
    doc = helpers.load_kats()
    mod = lowering.compile_module((helpers.REPO / doc["ir"]).read_text())
    for k in doc["kats"]:
        ins = [_vec(v) for v in k["inputs"]]
        want = _vec(k["expected"])
        got = mod.call(k["symbol"], *ins)                    # host buffers: staged, malloc'ed result
        assert isinstance(got, np.ndarray) and bits_equal(got, want), k["name"] + "\n" + mismatch_report(got, want)
        dev = mod.call(k["symbol"], *[torch.from_numpy(v).cuda() for v in ins])   # device buffers: in place
        assert dev.is_cuda and bits_equal(dev.cpu().numpy(), want), k["name"] + " (device)"


def test_outlined_stencil_part_of_a_solver_function_known_answer(env):
    """tests/golden/kat_outline_1d.mlir: @entry holds an implicit time_advance and is not lowered; the value it hands to
    the solver op is exported as entry__stencil_0(out, in) -- on the reference driver's input 1..16 that is KAT-2
    (smoke_time_advance.mlir:59-70), bit for bit, with host buffers and with device buffers"""
    lowering, torch = env
    mod = lowering.compile_module((helpers.GOLDEN_DIR / "kat_outline_1d.mlir").read_text())
    assert "entry" not in mod.symbols and "entry__stencil_0" in mod.symbols
    kat = [k for k in helpers.load_kats()["kats"] if k["symbol"] == "kat_react"][0]
    u, want = _vec(kat["inputs"][0]), _vec(kat["expected"])
    out = np.zeros(16)
    got = mod.call("entry__stencil_0", out, u)               # host buffers: a fresh malloc'ed result, @entry's arguments untouched
    assert isinstance(got, np.ndarray) and got is not out and bits_equal(got, want), mismatch_report(got, want)
    assert not out.any() and bits_equal(u, _vec(kat["inputs"][0]))
    dev = mod.call("entry__stencil_0", torch.zeros(16, dtype=torch.float64, device="cuda"), torch.from_numpy(u).cuda())
    assert dev.is_cuda and bits_equal(dev.cpu().numpy(), want)


@pytest.mark.parametrize("kind,shape", [("2d5", (33, 256)), ("3d7", (12, 10, 128)), ("3d27", (9, 8, 256)), ("3d7", (7, 6, 9))])
def test_fixture_entry_host_and_device(env, kind, shape):
    lowering, torch = env
    dt = np.float32 if kind == "3d27" else np.float64
    text = helpers.stencil_module(kind, shape)
    mod = lowering.compile_module(text)
    u = helpers.hash_field(shape, dt, seed=21)
    want = helpers.oracle_entry(kind, u)
    out = np.full_like(u, 5.0)
    res = mod.call("entry", out, u)                          # host path: H2D, kernel, D2H into `out`
    assert res is out and bits_equal(out, want), mismatch_report(out, want)
    d_in, d_out = torch.from_numpy(u).cuda(), torch.zeros(shape, dtype=torch.from_numpy(u).dtype, device="cuda")
    res = mod.call("entry", d_out, d_in)                     # device path: nothing crosses PCIe
    assert res is d_out and bits_equal(d_out.cpu().numpy(), want)
    opname = helpers.KINDS[kind][2]
    fresh = mod.call(opname, u)                              # opdef: callee-allocated result
    assert fresh is not u and bits_equal(fresh, want)
    kern = {a["function"]: a["kernel"] for a in mod.report["applies"]}
    assert kern[opname] == "march"


def test_in_place_update_keeps_reference_semantics(env):
    """store apply(load f) to f: the reference materialises the apply result before the store copies
    it (DataflowLowering.cpp:281 + :176-179), so passing the same buffer as source and destination is
    well defined; the lowering must not let the kernel write into a field it is still reading"""
    lowering, torch = env
    shape = (10, 12, 128)
    mod = lowering.compile_module(helpers.stencil_module("3d7", shape))
    u = helpers.hash_field(shape, np.float64, seed=4)
    want = helpers.oracle_entry("3d7", u)
    buf = u.copy()
    mod.call("entry", buf, buf)
    assert bits_equal(buf, want), mismatch_report(buf, want)
    d = torch.from_numpy(u).cuda()
    mod.call("entry", d, d)
    assert bits_equal(d.cpu().numpy(), want)


HEAT = '''
#l = #neptune_ir.location<"cell">
#b = #neptune_ir.bounds<lb = [0, 0, 0], ub = [{n0}, {n1}, {n2}]>
!t = !neptune_ir.temp<element = f64, bounds = #b, location = #l>
!f = !neptune_ir.field<element = f64, bounds = #b, location = #l>
module {{
  // variable-coefficient heat step: u + dt * k * lap(u), with a position-dependent source term
  neptune_ir.nonlinear_opdef @heat : (!t, !t) -> !t {{
  ^bb0(%u: !t, %k: !t):
    %r = neptune_ir.apply(%u, %k) attributes {{bounds = #neptune_ir.bounds<lb = [1, 1, 1], ub = [{m0}, {m1}, {m2}]>}}
      : (!t, !t) -> !t {{
      ^bb0(%i: index, %j: index, %kk: index, %a: !t, %c: !t):
        %c0 = neptune_ir.access %a[0, 0, 0] : !t -> f64
        %xm = neptune_ir.access %a[-1, 0, 0] : !t -> f64
        %xp = neptune_ir.access %a[1, 0, 0] : !t -> f64
        %ym = neptune_ir.access %a[0, -1, 0] : !t -> f64
        %yp = neptune_ir.access %a[0, 1, 0] : !t -> f64
        %zm = neptune_ir.access %a[0, 0, -1] : !t -> f64
        %zp = neptune_ir.access %a[0, 0, 1] : !t -> f64
        %kc = neptune_ir.access %c[0, 0, 0] : !t -> f64
        %six = arith.constant 6.0 : f64
        %dt = arith.constant 1.0e-3 : f64
        %s0 = arith.addf %xm, %xp : f64
        %s1 = arith.addf %s0, %ym : f64
        %s2 = arith.addf %s1, %yp : f64
        %s3 = arith.addf %s2, %zm : f64
        %s4 = arith.addf %s3, %zp : f64
        %s5 = arith.mulf %six, %c0 : f64
        %lap = arith.subf %s4, %s5 : f64
        %flux = arith.mulf %kc, %lap : f64
        %ij = arith.addi %i, %j : index
        %ijk = arith.muli %ij, %kk : index
        %w = arith.index_cast %ijk : index to i64
        %wf = arith.sitofp %w : i64 to f64
        %src = arith.divf %wf, %six : f64
        %rhs = arith.addf %flux, %src : f64
        %d = arith.mulf %dt, %rhs : f64
        %o = arith.addf %c0, %d : f64
        neptune_ir.yield %o : f64
      }}
    neptune_ir.return %r : !t
  }}
  func.func @step(%dst: memref<?x?x?xf64>, %src: memref<?x?x?xf64>, %coef: memref<?x?x?xf64>) -> memref<?x?x?xf64> {{
    %fd = neptune_ir.wrap %dst : memref<?x?x?xf64> -> !f
    %fs = neptune_ir.wrap %src : memref<?x?x?xf64> -> !f
    %fk = neptune_ir.wrap %coef : memref<?x?x?xf64> -> !f
    %u = neptune_ir.load %fs : !f -> !t
    %k = neptune_ir.load %fk : !f -> !t
    %y = neptune_ir.apply_nonlinear @heat(%u, %k) : (!t, !t) -> !t
    neptune_ir.store %y to %fd {{bounds = #neptune_ir.bounds<lb = [2, 0, 4], ub = [{m0}, {n1}, {m2}]>}} : !t to !f
    %res = neptune_ir.unwrap %fd : !f -> memref<?x?x?xf64>
    func.return %res : memref<?x?x?xf64>
  }}
}}
'''


@pytest.mark.parametrize("shape", [(9, 11, 128), (6, 5, 14)])
def test_two_input_body_with_index_arguments_and_bounded_store(env, shape):
    """march kernel with one halo input + one offset-0 input + region index arguments (wide rows),
    direct kernel for the narrow shape; neptune_ir.store with bounds"""
    lowering, torch = env
    text = HEAT.format(n0=shape[0], n1=shape[1], n2=shape[2], m0=shape[0] - 1, m1=shape[1] - 1, m2=shape[2] - 1)
    m = oracle.Module.parse(text)
    u = helpers.hash_field(shape, np.float64, seed=8)
    k = helpers.hash_field(shape, np.float64, seed=9) * 0.25 + 1.0
    want = np.full(shape, -1.0)
    m.call("step", want, u, k)
    mod = lowering.compile_module(text)
    got = np.full(shape, -1.0)
    mod.call("step", got, u, k)
    assert bits_equal(got, want), mismatch_report(got, want)
    kern = {a["function"]: a["kernel"] for a in mod.report["applies"]}
    assert kern["heat"] == "march"          # the template is march-capable; narrow rows fall back at run time
    fresh = mod.call("heat", torch.from_numpy(u).cuda(), torch.from_numpy(k).cuda())
    assert bits_equal(fresh.cpu().numpy(), m.call("heat", u, k))


def test_out_of_bounds_apply_aborts_like_the_reference_runtime_convention(env, tmp_path):
    """unconditional access outside the input box: undefined behaviour in the reference, refused
    here with the runtime's print-and-abort convention ([NeptuneRT] ... + abort())"""
    lowering, _ = env
    text = (helpers.GOLDEN_DIR / "kat_smoke_1d.mlir").read_text().replace(
        "bounds = #neptune_ir.bounds<lb = [1], ub = [15]>} : (!t16) -> !t16 {\n    ^bb0(%i: index, %a: !t16):\n      %m",
        "bounds = #neptune_ir.bounds<lb = [0], ub = [16]>} : (!t16) -> !t16 {\n    ^bb0(%i: index, %a: !t16):\n      %m", 1)
    script = tmp_path / "oob.py"
    script.write_text(f'''
import sys, numpy as np
sys.path.insert(0, {str(helpers.REPO / "neptune-pde-solver_amd")!r})
from neptune_hip import lowering
mod = lowering.compile_module({text!r})
mod.call("kat_lap", np.arange(1, 17, dtype=np.float64))
print("NOT REACHED")
''')
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True)
    assert p.returncode != 0 and "NOT REACHED" not in p.stdout
    assert "[NeptuneRT][HIP] kat_lap: neptune_ir.apply reads outside an input's bounds" in p.stderr


def test_python_dsl_jit_class_end_to_end(env):
    """reference-style user program: operator defined with @linear_op_def, solver class traced by
    @jit_class, called with NumPy arrays (host path) and CUDA tensors (device path)"""
    lowering, torch = env
    import neptune as nep
    n0, n1 = 40, 256
    box = ([0, 0], [n0, n1])

    @nep.jit_class
    class Heat2D:
        def __init__(self, alpha):
            self.alpha = alpha

        def define_operators(self):
            alpha = self.alpha

            @nep.linear_op_def(bounds=box, location="cell", apply_bounds=([1, 1], [n0 - 1, n1 - 1]))
            def lap2d(u):
                return (u[-1, 0] + u[1, 0] + u[0, -1] + u[0, 1] - 4.0 * u[0, 0]) * alpha

        def step(self, out, u):
            fout, fin = nep.wrap(out, box), nep.wrap(u, box)
            y = nep.apply_linear("lap2d", nep.load(fin))
            nep.store(y, fout)
            return nep.unwrap(fout)

    solver = Heat2D(0.125)
    u = helpers.hash_field((n0, n1), np.float64, seed=77)
    out = np.zeros_like(u)
    res = solver.step(out, u)
    want = u.copy()
    want[1:-1, 1:-1] = ((((u[:-2, 1:-1] + u[2:, 1:-1]) + u[1:-1, :-2]) + u[1:-1, 2:]) - 4.0 * u[1:-1, 1:-1]) * 0.125
    assert res is out and bits_equal(out, want), mismatch_report(out, want)
    d_out = torch.zeros((n0, n1), dtype=torch.float64, device="cuda")
    solver.step(d_out, torch.from_numpy(u).cuda())      # second call: no re-trace, device path
    assert bits_equal(d_out.cpu().numpy(), want)


def test_python_dsl_dot_product_and_time_step_on_the_device(env):
    """neptune.reduce_sum / neptune.time_advance (Python face of the fused kernels) built with the DSL, compiled through
    neptune.jit_compile and run on device-resident arrays"""
    lowering, torch = env
    import neptune as nep
    nep.reset()
    n0, n1 = 24, 256
    box = ([0, 0], [n0, n1])

    @nep.linear_op_def(bounds=box, location="cell", apply_bounds=([1, 1], [n0 - 1, n1 - 1]))
    def lap(u):
        return u[-1, 0] + u[1, 0] + u[0, -1] + u[0, 1] - 4.0 * u[0, 0]

    c = nep.get_compiler()
    c.start_function("dot", [("memref", 2), ("memref", 2)])
    a, b = (nep.load(nep.wrap(nep.Expr(c.get_function_arg(i)), box)) for i in range(2))

    @nep.apply(inputs=[a, b], bounds=box)
    def prod(x, y):
        return x[0, 0] * y[0, 0]

    c.create_return(nep.reduce_sum(prod)._handle)
    c.end_function()
    c.start_function("step", [("memref", 2), ("memref", 2)])
    fout, fin = nep.wrap(nep.Expr(c.get_function_arg(0)), box), nep.wrap(nep.Expr(c.get_function_arg(1)), box)
    nep.store(nep.time_advance(nep.load(fin), 0.125, lap), fout)
    c.create_return(nep.unwrap(fout)._handle)
    c.end_function()
    text = c.dump()
    mod = nep.jit_compile(c)
    nep.reset()
    m = oracle.Module.parse(text)
    u = helpers.hash_field((n0, n1), np.float64, seed=3)
    v = helpers.hash_field((n0, n1), np.float64, seed=4)
    du, dv = torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda()
    got = mod.call("dot", du, dv)
    want = float(m.call("dot", u, v))
    assert abs(got - want) <= 2 * u.size * np.finfo(np.float64).eps * float(np.abs(u * v).sum())
    out = torch.zeros_like(du)
    mod.call("step", out, du)
    ref = np.zeros_like(u)
    m.call("step", ref, u)
    assert bits_equal(out.cpu().numpy(), ref)


@pytest.mark.parametrize("elem", ["f64", "f32"])
def test_elementary_functions_within_a_few_ulp(env, elem):
    """bodies with math.exp / log / sin / cos / tanh / powf: device math library vs numpy (the oracle), stated tolerance
    8 ulp of the result's magnitude scale on every cell; march and direct kernels agree with each other bit for bit
    (same device functions), copy-through rim exact"""
    lowering, torch = env
    from test_lowering import ELEMENTARY
    n0, n1 = 20, 384
    text = ELEMENTARY.format(elem=elem, n0=n0, n1=n1, m0=n0 - 1, m1=n1 - 1)
    dt = np.float64 if elem == "f64" else np.float32
    mod = lowering.compile_module(text)
    assert mod.report["applies"][0]["exact"] is False
    u = helpers.hash_field((n0, n1), dt, seed=12)
    want = oracle.Module.parse(text).call("react", u)
    d_u = torch.from_numpy(u).cuda()
    got = mod.call("react", d_u).cpu().numpy()
    eps = float(np.finfo(dt).eps)
    scale = np.maximum(np.abs(want), 1.0)            # terms of magnitude <= ~3 are added: absolute errors of a few eps
    assert np.all(np.abs(got.astype(np.float64) - want.astype(np.float64)) <= 8 * eps * scale), \
        float(np.max(np.abs(got.astype(np.float64) - want) / (eps * scale)))
    assert bits_equal(got[0], u[0]) and bits_equal(got[:, -1], u[:, -1])
    os_env = __import__("os").environ
    os_env["NEPTUNE_HIP_KERNEL"] = "direct"
    try:
        got_direct = mod.call("react", d_u).cpu().numpy()
    finally:
        os_env.pop("NEPTUNE_HIP_KERNEL", None)
    assert bits_equal(got, got_direct)


def test_c_caller_in_the_style_of_the_petsc_matmult_thunk(env, tmp_path):
    """a plain C program looks the lowered operator up with dlsym(RTLD_DEFAULT), calls it with a host
    array and free()s the result -- the three things LinSolverCtx::MatMultThunk does
    (reference lib/Runtime/PETSc/NeptunePETScRuntime.cpp:182-230)"""
    lowering, _ = env
    doc = helpers.load_kats()
    mod = lowering.compile_module((helpers.REPO / doc["ir"]).read_text())
    exe = tmp_path / "thunk"
    subprocess.run(["gcc", "-O1", str(helpers.REPO / "tests/thunk_abi/matmult_thunk.c"), "-ldl", "-o", str(exe)], check=True)
    from neptune_hip import _capi
    env_vars = dict(__import__("os").environ, LD_LIBRARY_PATH=str(_capi.LIB_PATH.parent))
    for sym, n, name in (("kat_lap", 16, "KAT-1 ac_lap, driver input"), ("kat_bs", 32, "KAT-3 Black-Scholes operator")):
        p = subprocess.run([str(exe), str(mod.path), sym, str(n)], capture_output=True, text=True, env=env_vars)
        assert p.returncode == 0, p.stderr
        got = np.array([float.fromhex(t) for t in p.stdout.split()])
        want = _vec(next(k for k in doc["kats"] if k["name"] == name)["expected"])
        assert bits_equal(got, want), name


@pytest.mark.parametrize("fixture,shape,fused", [("explicit-heat-2d.mlir", (12, 128), True),
                                                 ("explicit-twostage-3d.mlir", (10, 9, 128), False)])
def test_explicit_time_advance_steps(env, fixture, shape, fused):
    """neptune_ir.time_advance {method = 0, rhs = @opdef}: u + dt * rhs(u), chained for several steps on
    device-resident fields (nothing crosses PCIe between steps), bit-exact against the oracle.  A
    single-apply rhs runs fused with the axpy (one kernel, two field passes); a two-apply rhs takes the
    call + axpy form."""
    lowering, torch = env
    text = (helpers.REPO / "tests/mlir_tests/time_stepping" / fixture).read_text()
    mod = lowering.compile_module(text)
    assert [a["inputs"] for a in mod.report["applies"] if a["function"] == "step"] == [1 if fused else 2]
    m = oracle.Module.parse(text)
    u = helpers.hash_field(shape, np.float64, seed=11)
    a, b = torch.from_numpy(u).cuda(), torch.zeros(shape, dtype=torch.float64, device="cuda")
    ha, hb = u.copy(), np.zeros_like(u)
    for _ in range(6):
        mod.call("step", b, a)
        m.call("step", hb, ha)
        a, b, ha, hb = b, a, hb, ha
    assert bits_equal(a.cpu().numpy(), ha), mismatch_report(a.cpu().numpy(), ha)
    out = np.zeros_like(u)
    mod.call("step", out, u)                              # host buffers
    want = np.zeros_like(u)
    m.call("step", want, u)
    assert bits_equal(out, want)
    # in place (state field == destination field): the apply may not write where it still reads
    d = torch.from_numpy(u).cuda()
    mod.call("step", d, d)
    assert bits_equal(d.cpu().numpy(), want)


def test_fused_time_step_on_every_kernel_and_tile(env, monkeypatch):
    """the fused step through the direct kernel and each default march tile, with chunk seams, on a
    3-D field whose rhs bounds leave a rim (rim cells: u + dt*u)"""
    lowering, torch = env
    text = (helpers.FIXTURE_DIR / "apply-3d-7pt.mlir").read_text()
    text = text.replace("ub = [512, 512, 512]", "ub = [11, 14, 256]").replace("ub = [511, 511, 511]", "ub = [10, 12, 255]")
    text = text.replace("lb = [1, 1, 1]", "lb = [2, 1, 1]")
    head, _, _ = text.partition("  func.func @entry")
    text = head + """  func.func @step(%out: memref<?x?x?xf64>, %in: memref<?x?x?xf64>) -> memref<?x?x?xf64> {
    %fo = neptune_ir.wrap %out : memref<?x?x?xf64> -> !field
    %fi = neptune_ir.wrap %in : memref<?x?x?xf64> -> !field
    %u0 = neptune_ir.load %fi : !field -> !temp
    %dt = arith.constant 2.5e-1 : f64
    %u1 = neptune_ir.time_advance %u0, %dt {method = 0 : i32, rhs = @lap3d} : !temp, f64 -> !temp
    neptune_ir.store %u1 to %fo : !temp to !field
    %res = neptune_ir.unwrap %fo : !field -> memref<?x?x?xf64>
    func.return %res : memref<?x?x?xf64>
  }
}
"""
    shape = (11, 14, 256)
    u = helpers.hash_field(shape, np.float64, seed=21)
    want = np.zeros_like(u)
    oracle.Module.parse(text).call("step", want, u)
    assert want[0, 3, 7] == u[0, 3, 7] + 0.25 * u[0, 3, 7]
    mod = lowering.compile_module(text)
    assert "step_ta0" in [a["tag"] for a in mod.report["applies"]]
    d_in = torch.from_numpy(u).cuda()
    for s in [{}, {"NEPTUNE_HIP_KERNEL": "direct"}] + [{"NEPTUNE_HIP_VARIANT": str(v), "NEPTUNE_HIP_CHUNK": "3"} for v in range(7)]:
        for k in ("NEPTUNE_HIP_KERNEL", "NEPTUNE_HIP_VARIANT", "NEPTUNE_HIP_CHUNK"):
            monkeypatch.delenv(k, raising=False)
        for k, v in s.items():
            monkeypatch.setenv(k, v)
        d_out = torch.zeros(shape, dtype=torch.float64, device="cuda")
        mod.call("step", d_out, d_in)
        got = d_out.cpu().numpy()
        assert bits_equal(got, want), f"{s}: " + mismatch_report(got, want)


def test_geometry_level_entry_tiles_regions_like_the_builtin_bodies(env):
    """LoweredModule.geom_entry: a user stencil launched region by region on caller-owned fields (what the slab
    decomposition does), equal to the module's own whole-field call; bad geometry is refused, not run"""
    lowering, torch = env
    from neptune_hip import _capi, apply, fields
    text = (helpers.FIXTURE_DIR / "apply-3d-13pt.mlir").read_text()
    mod = lowering.compile_module(text)
    entry = mod.geom_entry("lap13")
    shape = (20, 18, 256)
    u = helpers.hash_field(shape, np.float64, seed=19)
    want = mod.call("lap13", torch.from_numpy(u).cuda()).cpu().numpy()
    assert bits_equal(want, oracle.Module.parse(text).call("lap13", u))
    fin = fields.DeviceField.from_numpy(u)
    fout = fields.DeviceField.empty_like(fin)
    fout.tensor.fill_(-5.0)
    bounds = ([2, 2, 2], [18, 16, 254])
    full = lambda lo, hi: ([lo, 0, 0], [hi, shape[1], shape[2]])
    for kern in (_capi.KERNEL_MARCH, _capi.KERNEL_DIRECT):
        cfg = apply.make_cfg(kern)
        for lo, hi in ((0, 3), (17, 20), (3, 17)):
            apply.apply_builtin(entry, [fin], fout, bounds, region=full(lo, hi), cfg=cfg)
        torch.cuda.synchronize()
        assert bits_equal(fout.numpy(), want), mismatch_report(fout.numpy(), want)
        fout.tensor.fill_(-5.0)
    with pytest.raises(_capi.NeptuneHipError, match="EOOB"):      # bounds reaching the rim: reads outside the input
        apply.apply_builtin(entry, [fin], fout, ([1, 2, 2], [18, 16, 254]))


def test_step_loop_over_a_lowered_apply(env):
    """neptune_hip_step_loop with a lowered module's geometry-level entry (function pointer) in place of a built-in
    body: hipGraph-replayed ping-pong steps == the module's own @entry called step by step"""
    lowering, torch = env
    from neptune_hip import apply, fields
    sys.path.insert(0, str(helpers.REPO / "tools"))
    import make_stencil_mlir
    shape = (48, 256)
    text = make_stencil_mlir.stencil_module("2d5", list(shape))
    mod = lowering.compile_module(text)
    entry = mod.geom_entry("lap2d")
    u = helpers.hash_field(shape, np.float64, seed=27) * 0.125
    steps = 25
    a, b = torch.from_numpy(u).cuda(), torch.zeros(shape, dtype=torch.float64, device="cuda")
    for _ in range(steps):
        mod.call("entry", b, a)
        a, b = b, a
    want = a.cpu().numpy()
    fa = fields.DeviceField.from_numpy(u)
    fb = fields.DeviceField.empty_like(fa)
    last = apply.step_loop(entry, fa, fb, ([1, 1], [shape[0] - 1, shape[1] - 1]), steps)
    torch.cuda.synchronize()
    assert bits_equal(last.numpy(), want), mismatch_report(last.numpy(), want)


def test_first_use_tuning_keeps_the_bits(env, tmp_path):
    """NEPTUNE_HIP_TUNE=1: the first launch of each (apply, geometry) times the module's tiles and keeps the fastest;
    results stay bit-exact (every tile computes the same bits), later calls reuse the choice"""
    script = tmp_path / "tune.py"
    script.write_text(f"""
import os, sys
sys.path.insert(0, {str(helpers.REPO / 'neptune-pde-solver_amd')!r}); sys.path.insert(0, {str(helpers.REPO / 'tests')!r})
os.environ["NEPTUNE_CACHE_DIR"] = {str(tmp_path)!r}
os.environ["NEPTUNE_HIP_TUNE"] = "1"
import numpy as np, torch, helpers
from helpers import oracle, bits_equal
from neptune_hip import lowering, _capi, apply, fields
sys.path.insert(0, {str(helpers.REPO / 'tools')!r})
import make_stencil_mlir
cases = [(make_stencil_mlir.stencil_module("3d7", [20, 18, 256]), (20, 18, 256)),
         (make_stencil_mlir.stencil_module("2d5", [40, 512]), (40, 512)),
         ((helpers.FIXTURE_DIR / "apply-3d-13pt.mlir").read_text(), (20, 18, 256))]
for fixture, (text, shape) in enumerate(cases):
    mod = lowering.compile_module(text)
    u = helpers.hash_field(shape, np.float64, seed=33)
    want = np.zeros_like(u)
    oracle.Module.parse(text).call("entry", want, u)
    d_in = torch.from_numpy(u).cuda()
    for rep in range(3):
        d_out = torch.zeros(shape, dtype=torch.float64, device="cuda")
        mod.call("entry", d_out, d_in)
        assert bits_equal(d_out.cpu().numpy(), want), (fixture, rep)
# the library's built-in bodies take the same route when no configuration is given
a = fields.DeviceField.hashed((24, 16, 256), _capi.F64, seed=2)
b = fields.DeviceField.empty_like(a); c = fields.DeviceField.empty_like(a)
bounds = ([1, 1, 1], [23, 15, 255])
apply.apply_builtin(_capi.BODY_LAP3D7_F64, [a], b, bounds)
apply.apply_builtin(_capi.BODY_LAP3D7_F64, [a], c, bounds, cfg=apply.make_cfg(_capi.KERNEL_DIRECT))
torch.cuda.synchronize()
assert apply.count_mismatch(b, c) == 0
print("TUNE_OK")
""")
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "TUNE_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_launch_wisdom_is_measured_once_and_reused_by_the_next_process(env, tmp_path):
    """include/neptune_hip.h "launch wisdom": without any environment switch, the first launch of a large enough apply
    measures its tiles once and appends the choice to the wisdom file; a SECOND PROCESS finds it there and launches the
    same configuration without timing anything (tune_stats: 0 measured, hits > 0).  A lowered module and a built-in
    body; NEPTUNE_HIP_TUNE_MIN_CELLS lowered so that a small field counts as large.  Bits stay exact."""
    script = tmp_path / "wisdom.py"
    script.write_text(f"""
import ctypes as C, json, os, sys
sys.path.insert(0, {str(helpers.REPO / 'neptune-pde-solver_amd')!r}); sys.path.insert(0, {str(helpers.REPO / 'tests')!r})
os.environ["NEPTUNE_CACHE_DIR"] = {str(tmp_path)!r}
os.environ["NEPTUNE_HIP_TUNE_MIN_CELLS"] = "50000"
os.environ.pop("NEPTUNE_HIP_TUNE", None); os.environ.pop("NEPTUNE_HIP_WISDOM", None)
import numpy as np, torch, helpers
from helpers import oracle, bits_equal
from neptune_hip import lowering, _capi, apply, fields
lib = _capi.load()
shape = (24, 20, 256)
text = helpers.stencil_module("3d7", list(shape))
mod = lowering.compile_module(text)
u = helpers.hash_field(shape, np.float64, seed=33)
want = np.zeros_like(u)
oracle.Module.parse(text).call("entry", want, u)
d_in = torch.from_numpy(u).cuda()
last = []
for rep in range(3):
    d_out = torch.zeros(shape, dtype=torch.float64, device="cuda")
    mod.call("entry", d_out, d_in)
    assert bits_equal(d_out.cpu().numpy(), want)
    cfg = _capi.LaunchCfg()
    assert lib.neptune_hip_last_launch(C.byref(cfg)) == 1
    last.append((cfg.kernel, cfg.variant, cfg.chunk))
assert last[0] == last[1] == last[2] and last[0][0] == _capi.KERNEL_MARCH and last[0][2] > 0, last
a = fields.DeviceField.hashed(shape, _capi.F64, seed=2)
b = fields.DeviceField.empty_like(a); c = fields.DeviceField.empty_like(a)
bounds = ([1, 1, 1], [n - 1 for n in shape])
apply.apply_builtin(_capi.BODY_LAP3D7_F64, [a], b, bounds)
cfg = _capi.LaunchCfg(); lib.neptune_hip_last_launch(C.byref(cfg))
apply.apply_builtin(_capi.BODY_LAP3D7_F64, [a], c, bounds, cfg=apply.make_cfg(_capi.KERNEL_DIRECT))
torch.cuda.synchronize()
assert apply.count_mismatch(b, c) == 0
# a small field is left alone: the fixed automatic tile, nothing measured
small = fields.DeviceField.hashed((8, 8, 128), _capi.F64, seed=3)
apply.apply_builtin(_capi.BODY_LAP3D7_F64, [small], fields.DeviceField.empty_like(small), ([1, 1, 1], [7, 7, 127]))
stats = (C.c_int64 * 3)()
lib.neptune_hip_tune_stats(stats)
print("WISDOM", json.dumps({{"measured": stats[0], "hits": stats[1], "stored": stats[2], "module": last[0],
                             "builtin": (cfg.kernel, cfg.variant, cfg.chunk), "path": lib.neptune_hip_wisdom_path().decode()}}))
""")
    import json
    runs = []
    for _ in range(2):
        p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900)
        assert p.returncode == 0 and "WISDOM" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
        runs.append(json.loads(p.stdout.split("WISDOM", 1)[1].strip().splitlines()[0]))
    first, second = runs
    assert first["path"] == str(tmp_path / "wisdom_v1.txt") and (tmp_path / "wisdom_v1.txt").exists()
    assert first["measured"] == 2 and first["stored"] == 2 and first["hits"] == 0, first      # the module's apply and the built-in body
    assert second["measured"] == 0 and second["stored"] == 0 and second["hits"] == 2, second  # nothing timed: both found in the file
    assert second["module"] == first["module"] and second["builtin"] == first["builtin"]
    assert len((tmp_path / "wisdom_v1.txt").read_text().splitlines()) == 2


def test_async_mode_keeps_results_and_order(env, tmp_path):
    """NEPTUNE_HIP_ASYNC=1: device-resident calls return without synchronising; chained steps (pooled temporaries reused
    while earlier kernels may still run), a device result and a scalar result still come out right"""
    script = tmp_path / "async_mode.py"
    script.write_text(f"""
import os, sys
sys.path.insert(0, {str(helpers.REPO / 'neptune-pde-solver_amd')!r}); sys.path.insert(0, {str(helpers.REPO / 'tests')!r})
os.environ["NEPTUNE_CACHE_DIR"] = {str(tmp_path)!r}
os.environ["NEPTUNE_HIP_ASYNC"] = "1"
import numpy as np, torch, helpers
from helpers import oracle, bits_equal
from neptune_hip import lowering
text = (helpers.REPO / "tests/mlir_tests/time_stepping/explicit-twostage-3d.mlir").read_text()
mod = lowering.compile_module(text)
m = oracle.Module.parse(text)
u = helpers.hash_field((10, 9, 128), np.float64, seed=11)
a, b = torch.from_numpy(u).cuda(), torch.zeros((10, 9, 128), dtype=torch.float64, device="cuda")
ha, hb = u.copy(), np.zeros_like(u)
for _ in range(40):
    mod.call("step", b, a)
    a, b = b, a
for _ in range(40):
    m.call("step", hb, ha)
    ha, hb = hb, ha
assert bits_equal(a.cpu().numpy(), ha)
fresh = mod.call("rhs", torch.from_numpy(u).cuda())                    # callee-allocated device result
assert bits_equal(fresh.cpu().numpy(), m.call("rhs", u))
out = np.zeros_like(u)
mod.call("step", out, u)                                               # host buffers: still synchronous
want = np.zeros_like(u); m.call("step", want, u)
assert bits_equal(out, want)
print("ASYNC_OK")
""")
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "ASYNC_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_temporaries_come_from_the_block_pool(env):
    """an apply result that cannot be written into a destination field (here: in-place update, and the rhs
    temp of the two-stage step) is a pooled device block: cached when the call returns, reused by the next
    call, released by neptune_hip_pool_trim"""
    lowering, torch = env
    from neptune_hip import _capi
    lib = _capi.load()
    text = (helpers.REPO / "tests/mlir_tests/time_stepping/explicit-twostage-3d.mlir").read_text()
    mod = lowering.compile_module(text)
    u = helpers.hash_field((10, 9, 128), np.float64, seed=4)
    want = np.zeros_like(u)
    oracle.Module.parse(text).call("step", want, u)
    lib.neptune_hip_pool_trim()
    assert lib.neptune_hip_pool_cached_bytes() == 0
    a, b = torch.from_numpy(u).cuda(), torch.zeros(u.shape, dtype=torch.float64, device="cuda")
    mod.call("step", b, a)
    cached = lib.neptune_hip_pool_cached_bytes()
    assert cached >= 2 * u.nbytes            # the two temps of @rhs
    for _ in range(3):
        b.zero_()
        mod.call("step", b, a)
        assert lib.neptune_hip_pool_cached_bytes() == cached      # reused, not grown
        assert bits_equal(b.cpu().numpy(), want)
    lib.neptune_hip_pool_trim()
    assert lib.neptune_hip_pool_cached_bytes() == 0
    mod.call("step", b, a)                   # and allocation from scratch still works
    assert bits_equal(b.cpu().numpy(), want)


def test_radius_two_star_stencil_on_the_march_kernel(env):
    """4th-order 13-point Laplacian: 5 live planes, 2-deep J and K halos (LDS exchange of two rows,
    two-cell wave shifts, two scalar halo cells per side)"""
    lowering, torch = env
    text = (helpers.FIXTURE_DIR / "apply-3d-13pt.mlir").read_text()
    mod = lowering.compile_module(text)
    assert {a["function"]: a["kernel"] for a in mod.report["applies"]}["lap13"] == "march"
    u = helpers.hash_field((20, 18, 256), np.float64, seed=13)
    want = np.zeros_like(u)
    oracle.Module.parse(text).call("entry", want, u)
    d_out = torch.zeros(u.shape, dtype=torch.float64, device="cuda")
    mod.call("entry", d_out, torch.from_numpy(u).cuda())
    got = d_out.cpu().numpy()
    assert bits_equal(got, want), mismatch_report(got, want)
    assert bits_equal(got[:2], u[:2]) and bits_equal(got[:, :, -2:], u[:, :, -2:])   # 2-cell copy-through rim


def test_smoke_script_in_the_shape_of_the_reference_smoke_apply(tmp_path):
    """tests/smoke_tests/smoke_apply_hip.sh: lower, build, generate the reference's driver, link, run"""
    p = subprocess.run(["bash", str(helpers.REPO / "tests/smoke_tests/smoke_apply_hip.sh")], capture_output=True, text=True,
                       env=dict(__import__("os").environ, WORKDIR=str(tmp_path)))
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    assert "SMOKE_OK" in p.stdout and "x[1]=0.000000" in p.stdout and "x[23]=24.000000" in p.stdout


ZERO_TRIP = '''
#l = #neptune_ir.location<"cell">
#b = #neptune_ir.bounds<lb = [0, 0], ub = [12, 256]>
!t = !neptune_ir.temp<element = f64, bounds = #b, location = #l>
!f = !neptune_ir.field<element = f64, bounds = #b, location = #l>
module {
  func.func @entry(%o: memref<?x?xf64>, %a: memref<?x?xf64>) -> f64 {
    %fa = neptune_ir.wrap %a : memref<?x?xf64> -> !f
    %fo = neptune_ir.wrap %o : memref<?x?xf64> -> !f
    %u = neptune_ir.load %fa : !f -> !t
    %r = neptune_ir.apply(%u) attributes {bounds = #neptune_ir.bounds<lb = [7, 1], ub = [5, 255]>} : (!t) -> !t {
      ^bb0(%i: index, %j: index, %x: !t):
        %w = neptune_ir.access %x[-1, 0] : !t -> f64
        %e = neptune_ir.access %x[1, 0] : !t -> f64
        %s = arith.addf %w, %e : f64
        neptune_ir.yield %s : f64
    }
    neptune_ir.store %r to %fo : !t to !f
    %z = neptune_ir.reduce %u in #neptune_ir.bounds<lb = [3, 9], ub = [9, 4]> {kind = "sum"} : !t -> f64
    func.return %z : f64
  }
}
'''


def test_loop_bounds_with_lb_above_ub_make_zero_trips(env):
    """an apply or reduce whose bounds run backwards along a dimension lowers to scf.for loops that make no
    trip in the reference (DataflowLowering.cpp:289-310, :604-611): nothing is computed, the sum is 0"""
    lowering, torch = env
    mod = lowering.compile_module(ZERO_TRIP)
    u = helpers.hash_field((12, 256), np.float64, seed=41)
    want = np.full_like(u, -3.0)
    want_sum = oracle.Module.parse(ZERO_TRIP).call("entry", want, u)
    d_out = torch.full(u.shape, -3.0, dtype=torch.float64, device="cuda")
    got_sum = mod.call("entry", d_out, torch.from_numpy(u).cuda())
    assert want_sum == 0.0 and got_sum == 0.0
    assert bits_equal(d_out.cpu().numpy(), want)

"""SURVEY 8(f) rows 1+2 together: a matrix-free CG whose operator (`neptune_ir.apply` behind `@matmult`) and dot
products (`reduce(apply(a*b))`, one fused kernel) are lowered from NeptuneIR and run on device-resident vectors.
The same driver runs on the oracle; the two solutions agree to solver tolerance (dot products are tree sums on
the device, serial sums in the oracle, so iterates are not bit-identical)."""
import sys

import numpy as np
import pytest

import helpers
from helpers import oracle

pytestmark = pytest.mark.gpu
sys.path.insert(0, str(helpers.REPO / "examples"))


def test_matrix_free_cg_on_device_matches_the_oracle_driven_solve(built_libs, tmp_path, monkeypatch):
    monkeypatch.setenv("NEPTUNE_CACHE_DIR", str(tmp_path))
    import torch
    import cg_matrix_free as ex
    from neptune_hip import lowering
    shape = (14, 12, 128)
    text = ex.module_text(shape)
    mod = lowering.compile_module(text)
    kern = {a["function"]: a["kernel"] for a in mod.report["applies"]}
    assert kern == {"A": "march", "dot": "reduce"}
    rng = np.random.default_rng(3)
    b = np.zeros(shape)
    b[1:-1, 1:-1, 1:-1] = rng.random(tuple(n - 2 for n in shape))
    # the operator alone: bit-exact
    m = oracle.Module.parse(text)
    y_ref = np.zeros(shape)
    m.call("matmult", y_ref, b)
    d_b = torch.from_numpy(b).cuda()
    d_y = torch.zeros_like(d_b)
    mod.call("matmult", d_y, d_b)
    assert helpers.bits_equal(d_y.cpu().numpy(), y_ref)
    # the solve
    x_ref, it_ref, rel_ref = ex.cg(lambda y, v: m.call("matmult", y, v), lambda u, v: float(m.call("dot", u, v)),
                                   b, np.zeros(shape), tol=1e-10)
    d_x, it, rel = ex.cg(lambda y, v: mod.call("matmult", y, v), lambda u, v: mod.call("dot", u, v),
                         d_b, torch.zeros_like(d_b), tol=1e-10)
    x = d_x.cpu().numpy()
    assert rel <= 1e-10 and abs(it - it_ref) <= 2
    assert np.abs(x - x_ref).max() <= 1e-8 * np.abs(x_ref).max()
    resid = np.zeros(shape)
    m.call("matmult", resid, x)
    assert np.abs(resid - b).max() <= 1e-8 * np.abs(b).max()
    assert np.all(x[0] == 0) and np.all(x[:, :, -1] == 0)          # rim untouched: A is the identity there, b = 0


def test_examples_run(built_libs, tmp_path):
    """the example programs end to end at small sizes (DSL -> jit -> lowered calls, hipGraph step loop, CG, the
    25-point leapfrog wave step, the staggered-grid
    projection with its face fields in boxes of their own)"""
    import os
    import subprocess
    env = dict(os.environ, NEPTUNE_CACHE_DIR=str(tmp_path))
    heat = subprocess.run([sys.executable, str(helpers.REPO / "examples/heat_step.py"), "256", "200"], env=env,
                          capture_output=True, text=True, timeout=600)
    assert heat.returncode == 0 and "results agree: True" in heat.stdout, heat.stdout[-1500:] + heat.stderr[-3000:]
    cg = subprocess.run([sys.executable, str(helpers.REPO / "examples/cg_matrix_free.py"), "64"], env=env,
                        capture_output=True, text=True, timeout=600)
    assert cg.returncode == 0 and "CG:" in cg.stdout and "('dot', 'reduce')" in cg.stdout, cg.stdout[-1500:] + cg.stderr[-3000:]
    wave = subprocess.run([sys.executable, str(helpers.REPO / "examples/wave_25pt.py"), "64", "20"], env=env,
                          capture_output=True, text=True, timeout=600)
    assert wave.returncode == 0 and "stable: True" in wave.stdout and "('step', 'march')" in wave.stdout, \
        wave.stdout[-1500:] + wave.stderr[-3000:]
    stag = subprocess.run([sys.executable, str(helpers.REPO / "examples/staggered_projection.py"), "200", "5"], env=env,
                          capture_output=True, text=True, timeout=600)
    assert stag.returncode == 0 and "results agree: True" in stag.stdout and "('project', 'march')" in stag.stdout, \
        stag.stdout[-1500:] + stag.stderr[-3000:]


def test_c_krylov_loop_with_device_pointers_through_the_reference_abi(built_libs, tmp_path, monkeypatch):
    """SURVEY 8(f) row 2 without PCIe in the loop: tests/thunk_abi/cg_device.c is a 50-iteration CG in C that reaches
    @matmult / @dot with dlsym and the expanded-memref ABI (NeptunePETScRuntime.cpp:182-230) but hands over DEVICE
    pointers; vector updates are neptune_hip_axpy / _xpay.  Checked against the same 50 iterations driven by the oracle
    (dot products: device tree sum vs serial sum, so iterates agree to rounding, not bit for bit), and the block pool
    must stay empty: a staged host argument would have left its device shadow there."""
    import os
    import subprocess
    monkeypatch.setenv("NEPTUNE_CACHE_DIR", str(tmp_path))
    import cg_matrix_free as ex
    from neptune_hip import _capi, lowering
    shape = (14, 12, 128)
    iters = 50
    text = ex.module_text(shape)
    mod = lowering.compile_module(text)
    rng = np.random.default_rng(11)
    b = np.zeros(shape)
    b[1:-1, 1:-1, 1:-1] = rng.random(tuple(n - 2 for n in shape))
    (tmp_path / "b.bin").write_bytes(b.tobytes())
    exe = tmp_path / "cg_device"
    libdir = _capi.LIB_PATH.parent
    subprocess.run(["gcc", "-O1", "-std=c11", "-Wall", "-Werror", "-I", str(helpers.REPO / "include"),
                    str(helpers.REPO / "tests/thunk_abi/cg_device.c"), "-L", str(libdir), "-lneptune_hip", "-ldl",
                    f"-Wl,-rpath,{libdir}", "-o", str(exe)], check=True)
    p = subprocess.run([str(exe), str(mod.path), *[str(n) for n in shape], str(iters), str(tmp_path / "b.bin"),
                        str(tmp_path / "x.bin")], capture_output=True, text=True, env=dict(os.environ), timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = p.stdout.strip().splitlines()
    rs_dev = [float.fromhex(l.split()[3]) for l in lines if l.startswith("it ")]
    assert len(rs_dev) == iters
    assert lines[-1] == "pool_cached_bytes 0"                      # nothing was staged through a device shadow
    x_dev = np.frombuffer((tmp_path / "x.bin").read_bytes(), dtype=np.float64).reshape(shape)
    # the same loop on the oracle (numpy: `alpha * p` rounds, then the sum rounds -- the two roundings of the kernels)
    m = oracle.Module.parse(text)
    x, r = np.zeros(shape), b.copy()
    pv, ap = r.copy(), np.zeros(shape)
    rs = float(m.call("dot", r, r))
    rs_ref = []
    for _ in range(iters):
        m.call("matmult", ap, pv)
        alpha = rs / float(m.call("dot", pv, ap))
        x += alpha * pv
        r += (-alpha) * ap
        rs_new = float(m.call("dot", r, r))
        pv = r + (rs_new / rs) * pv
        rs = rs_new
        rs_ref.append(rs)
    assert rs_dev[-1] < 1e-8 * rs_dev[0]                                      # it converges
    # early iterations agree to rounding; later ones drift apart as CG amplifies the summation-order differences
    np.testing.assert_allclose(rs_dev[:10], rs_ref[:10], rtol=1e-9)
    assert np.abs(x_dev - x).max() <= 1e-8 * np.abs(x).max()
    assert np.all(x_dev[0] == 0) and np.all(x_dev[:, :, -1] == 0)


@pytest.mark.parametrize("mode", ["host", "device"])
def test_c_caller_in_the_shape_of_the_snes_residual_thunk(built_libs, tmp_path, monkeypatch, mode):
    """tests/thunk_abi/snes_residual_thunk.c reaches a lowered nonlinear_opdef the way the reference's SNES callback does
    (NeptunePETScRuntime.cpp:1303-1361: dlsym, the iterate and TWO captures as expanded rank-2 memrefs, a NeptuneMemRef2D
    back by value, released by the caller): with host arrays and plain free() -- the reference's thunk unchanged -- and
    with device pointers and neptune_rt_free.  Bit for bit against the oracle (scf.if rim branch included)."""
    import os
    import subprocess
    monkeypatch.setenv("NEPTUNE_CACHE_DIR", str(tmp_path))
    from neptune_hip import _capi, lowering
    text = (helpers.REPO / "tests/mlir_tests/nonlinear/residual-2d-2cap.mlir").read_text()
    mod = lowering.compile_module(text)
    assert {a["function"]: (a["kernel"], a["inputs"]) for a in mod.report["applies"]}["residual"] == ("march", 3)
    shape = (48, 256)
    ins = [helpers.hash_field(shape, np.float64, seed=60 + k) for k in range(3)]
    want = oracle.Module.parse(text).call("residual", *ins)
    for k, a in enumerate(ins):
        (tmp_path / f"in{k}.bin").write_bytes(a.tobytes())
    exe = tmp_path / "snes_thunk"
    libdir = _capi.LIB_PATH.parent
    subprocess.run(["gcc", "-O1", "-std=c11", "-Wall", "-Werror", "-I", str(helpers.REPO / "include"),
                    str(helpers.REPO / "tests/thunk_abi/snes_residual_thunk.c"), "-L", str(libdir), "-lneptune_hip", "-ldl",
                    f"-Wl,-rpath,{libdir}", "-o", str(exe)], check=True)
    p = subprocess.run([str(exe), str(mod.path), "residual", "48", "256", *[str(tmp_path / f"in{k}.bin") for k in range(3)],
                        str(tmp_path / "F.bin"), mode], capture_output=True, text=True, env=dict(os.environ), timeout=600)
    assert p.returncode == 0 and f"SNES_THUNK_OK {mode}" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
    got = np.frombuffer((tmp_path / "F.bin").read_bytes(), dtype=np.float64).reshape(shape)
    assert helpers.bits_equal(got, want), helpers.mismatch_report(got, want)
    assert helpers.bits_equal(got[0], (ins[0] - ins[1])[0])            # the rim branch: x - up
    if mode == "device":
        assert "pool_cached_bytes 0" in p.stdout                       # nothing was staged through a device shadow


@pytest.mark.gpu
def test_vector_updates_match_numpy_for_every_length_and_alignment(built_libs):
    """neptune_hip_axpy / _xpay: y + a*x and x + a*y with two roundings (no FMA), 16-byte vector form on aligned buffers with
    the last n % VK elements through one lane, scalar form on misaligned ones -- bit for bit what numpy computes"""
    import numpy as np
    import torch
    from neptune_hip import _capi
    lib = _capi.load()
    lib.neptune_hip_init(0)
    rng = np.random.default_rng(9)
    for npdt, tdt, code in ((np.float64, torch.float64, _capi.F64), (np.float32, torch.float32, _capi.F32)):
        for n in (1, 2, 3, 4, 5, 7, 8, 255, 256, 257, 1023, 4099, 100003):
            for shift in (0, 1):
                x = rng.standard_normal(n + shift).astype(npdt)
                y = rng.standard_normal(n + shift).astype(npdt)
                a = npdt(0.37)
                dx, dy = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
                assert lib.neptune_hip_axpy(code, n, float(a), dx[shift:].data_ptr(), dy[shift:].data_ptr(), None) == 0
                want = y.copy()
                want[shift:] = y[shift:] + a * x[shift:]
                assert np.array_equal(dy.cpu().numpy().view(np.uint8), want.view(np.uint8)), (npdt, n, shift, "axpy")
                dy2 = torch.from_numpy(y).cuda()
                assert lib.neptune_hip_xpay(code, n, dx[shift:].data_ptr(), float(a), dy2[shift:].data_ptr(), None) == 0
                want2 = y.copy()
                want2[shift:] = x[shift:] + a * y[shift:]
                assert np.array_equal(dy2.cpu().numpy().view(np.uint8), want2.view(np.uint8)), (npdt, n, shift, "xpay")

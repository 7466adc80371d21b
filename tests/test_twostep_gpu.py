"""Two or three applies per pass over HBM (csrc/kernels/apply_march2.hpp, neptune_hip_apply_chain_builtin,
<tag>__geom2 / __geom3, neptune_hip_step_loop_chain): out = A(A(in)) or A(A(A(in))) computed in one launch must equal
separate launches bit for bit -- the same operations on the same operands -- and both must equal the oracle's chained applies."""
import ctypes as C

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nh():
    import torch
    from neptune_hip import _capi, apply, fields

    class NS:
        pass
    ns = NS()
    ns.torch, ns.capi, ns.apply, ns.fields = torch, _capi, apply, fields
    ns.lib = _capi.load()
    ns.lib.neptune_hip_init(0)
    return ns


# shapes: window seams along every axis (44 rows per workgroup of the default 3 x 16 window, 52 / 28 / 28 of the others;
# 120 columns per wave; chunk seams), first / last tiles partially outside the field, a shifted logical origin and bounds
# tighter than the interior
CASES = [
    ((9, 40, 256), None, None),
    ((7, 100, 256), None, None),
    ((34, 61, 376), None, None),
    ((20, 29, 128), None, None),                       # one wave span, rows just over one workgroup window
    ((12, 30, 256), [5, -3, 7], None),                 # shifted origin (index arguments are not used by this body, boxes are)
    ((16, 40, 256), None, ([2, 3, 9], [13, 33, 201])),  # bounds tighter than the interior: copy-through bands inside
]


@pytest.fixture(autouse=True)
def chains_at_every_size(monkeypatch):
    """step loops chain only fields of >= 4e6 cells by default (smaller ones stay in the memory-side cache and are faster
    one apply per launch); the parity cases here are small, so lift the threshold"""
    monkeypatch.setenv("NEPTUNE_HIP_CHAIN_MIN_CELLS", "0")


@pytest.fixture(params=["0", "1", "2", "3"], ids=["rows3x16", "rows7x8", "rows4x8", "rows2x16"])
def window(request, monkeypatch):
    """every window shape the library holds (NEPTUNE_HIP_MARCH2 is read once per process, at the first pair launch:
    the other shapes are reached through a child process)"""
    return request.param


def test_every_window_shape_in_a_child_process(window, built_libs):
    import os
    import subprocess
    import sys
    if window == "0":
        pytest.skip("the default window is what every other test of this file runs")
    env = dict(os.environ, NEPTUNE_HIP_MARCH2=window)
    p = subprocess.run([sys.executable, "-m", "pytest", __file__, "-x", "-q", "-m", "gpu", "-k", "one_pass_equal or step_loop_uses_pairs",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]


@pytest.mark.parametrize("applies", [2, 3])
@pytest.mark.parametrize("shape,origin,bounds", CASES)
def test_two_applies_in_one_pass_equal_two_launches_and_the_oracle(nh, shape, origin, bounds, applies):
    body = nh.capi.BODY_LAP3D7_F64
    lb = origin or [0, 0, 0]
    u = helpers.hash_field(shape, np.float64, seed=23)
    fin = nh.fields.DeviceField.from_numpy(u, lb)
    mid = nh.fields.DeviceField.empty_like(fin)
    two = nh.fields.DeviceField.empty_like(fin)
    one = nh.fields.DeviceField.empty_like(fin)
    one.tensor.fill_(float("nan"))
    if bounds is None:
        b = ([l + 1 for l in lb], [l + n - 1 for l, n in zip(lb, shape)])
    else:
        b = ([l + x for l, x in zip(lb, bounds[0])], [l + x for l, x in zip(lb, bounds[1])])
    nh.apply.apply_builtin(body, [fin], mid, b)
    nh.apply.apply_builtin(body, [mid], two, b)
    o = helpers.oracle_entry("3d7", helpers.oracle_entry("3d7", u, origin=lb, bounds=b), origin=lb, bounds=b)
    if applies == 3:                         # a third launch / oracle apply on top
        three = nh.fields.DeviceField.empty_like(fin)
        nh.apply.apply_builtin(body, [two], three, b)
        two = three
        o = helpers.oracle_entry("3d7", o, origin=lb, bounds=b)
    for chunk in (0, 5):
        one.tensor.fill_(float("nan"))
        assert nh.apply.apply_twice(body, fin, one, b, cfg=nh.apply.make_cfg(chunk=chunk), applies=applies)
        nh.torch.cuda.synchronize()
        got, want = one.numpy(), two.numpy()
        assert helpers.bits_equal(got, want), f"chunk {chunk}\n" + helpers.mismatch_report(got, want)
    assert helpers.bits_equal(two.numpy(), o), helpers.mismatch_report(two.numpy(), o)


def test_geometries_the_pair_kernel_cannot_take_are_refused_not_run(nh):
    body = nh.capi.BODY_LAP3D7_F64
    for shape in ((8, 16, 100), (8, 16, 12), (8, 4, 128)):   # rows not whole 64-byte granules / too short / too few rows
        fin = nh.fields.DeviceField.from_numpy(helpers.hash_field(shape, np.float64, seed=1))
        out = nh.fields.DeviceField.empty_like(fin)
        out.tensor.fill_(7.0)
        assert nh.apply.apply_twice(body, fin, out, ([1, 1, 1], [n - 1 for n in shape])) is False
        nh.torch.cuda.synchronize()
        assert bool((out.tensor == 7.0).all())
    # 2-D rows that are not whole granules, and box bodies: one launch per apply
    f2 = nh.fields.DeviceField.from_numpy(helpers.hash_field((64, 252), np.float64, seed=1))
    assert nh.apply.apply_twice(nh.capi.BODY_LAP2D5_F64, f2, nh.fields.DeviceField.empty_like(f2), ([1, 1], [63, 251])) is False
    f3 = nh.fields.DeviceField.from_numpy(helpers.hash_field((8, 16, 256), np.float32, seed=1))
    assert nh.apply.apply_twice(nh.capi.BODY_LAP3D27_F32, f3, nh.fields.DeviceField.empty_like(f3), ([1, 1, 1], [7, 15, 255])) is False


@pytest.mark.parametrize("steps", [3, 4, 5, 6, 7, 9, 12, 37, 60, 71])   # from 17 launches on, 16 are replayed as a graph
def test_step_loop_uses_pairs_and_ends_in_the_documented_field(nh, steps, monkeypatch):
    """neptune_hip_step_loop with pair launches == the same loop forced to single launches == the oracle's chain; the
    newest state is in fields[steps % 2] either way"""
    body = nh.capi.BODY_LAP3D7_F64
    shape = (12, 33, 256)
    u = helpers.hash_field(shape, np.float64, seed=4) * 0.05      # the iteration grows ~12x per step: keep it finite
    bounds = ([1, 1, 1], [n - 1 for n in shape])
    a = nh.fields.DeviceField.from_numpy(u)
    b = nh.fields.DeviceField.empty_like(a)
    res = nh.apply.step_loop(body, a, b, bounds, steps)
    nh.torch.cuda.synchronize()
    assert res is (b if steps % 2 else a)
    got = res.numpy().copy()
    for env in ("NEPTUNE_HIP_NO_TRIPLES", "NEPTUNE_HIP_NO_PAIRS", "default threshold"):   # two per pass only, then one per pass
        if env == "default threshold":
            monkeypatch.delenv("NEPTUNE_HIP_NO_TRIPLES")
            monkeypatch.delenv("NEPTUNE_HIP_NO_PAIRS")
            monkeypatch.delenv("NEPTUNE_HIP_CHAIN_MIN_CELLS")
        else:
            monkeypatch.setenv(env, "1")
        a2 = nh.fields.DeviceField.from_numpy(u)
        b2 = nh.fields.DeviceField.empty_like(a2)
        res2 = nh.apply.step_loop(body, a2, b2, bounds, steps)
        nh.torch.cuda.synchronize()
        assert res2 is (b2 if steps % 2 else a2)
        assert helpers.bits_equal(got, res2.numpy()), env
    o = u
    for _ in range(steps):
        o = helpers.oracle_entry("3d7", o)
    assert helpers.bits_equal(got, o), helpers.mismatch_report(got, o)


def test_fused_euler_step_of_a_lowered_module_two_steps_per_pass(nh, tmp_path, monkeypatch):
    """the lowered @step (time_advance fused with its rhs apply) through its pair entry step_ta0__geom2, against the
    oracle running @step twice; then a whole step loop over the module's entries"""
    monkeypatch.setenv("NEPTUNE_CACHE_DIR", str(tmp_path))
    from neptune_hip import lowering
    shape = (10, 36, 256)
    text = helpers.stencil_module("3d7", list(shape), time_step=0.125)
    mod = lowering.compile_module(text)
    entry = mod.geom_entry("step")
    assert entry.symbol == "step_ta0__geom" and entry.fn2 is not None and entry.fn3 is not None
    u = helpers.hash_field(shape, np.float64, seed=8)
    m = helpers.oracle.Module.parse(text)
    o1, o2 = np.zeros(shape), np.zeros(shape)
    m.call("step", o1, u)
    m.call("step", o2, o1)
    bounds = ([1, 1, 1], [n - 1 for n in shape])
    fin = nh.fields.DeviceField.from_numpy(u)
    out = nh.fields.DeviceField.empty_like(fin)
    assert nh.apply.apply_twice(entry, fin, out, bounds)
    nh.torch.cuda.synchronize()
    assert helpers.bits_equal(out.numpy(), o2), helpers.mismatch_report(out.numpy(), o2)
    o3 = np.zeros(shape)
    m.call("step", o3, o2)
    assert nh.apply.apply_twice(entry, fin, out, bounds, applies=3)
    nh.torch.cuda.synchronize()
    assert helpers.bits_equal(out.numpy(), o3), helpers.mismatch_report(out.numpy(), o3)
    # the operator itself (lap3d_0__geom2) and a 6-step loop of the Euler step
    lap = mod.geom_entry("lap3d")
    assert lap.fn2 is not None
    a = nh.fields.DeviceField.from_numpy(u)
    b = nh.fields.DeviceField.empty_like(a)
    res = nh.apply.step_loop(entry, a, b, bounds, 6)
    nh.torch.cuda.synchronize()
    ha, hb = u.copy(), np.zeros(shape)
    for _ in range(6):
        m.call("step", hb, ha)
        ha, hb = hb, ha
    assert helpers.bits_equal(res.numpy(), ha), helpers.mismatch_report(res.numpy(), ha)


@pytest.mark.parametrize("elem", ["f32", "f64"])
def test_chains_of_a_generated_body_with_an_index_argument(nh, elem, tmp_path, monkeypatch):
    """fp32 takes other window constants (16 cells per store granule, 4 per lane), and a body that reads an index
    argument sees a different plane / cell coordinate in every stage: a generated radius-1 star (weighted sum plus the
    last index) through __geom2 / __geom3 against repeated single launches and the oracle"""
    monkeypatch.setenv("NEPTUNE_CACHE_DIR", str(tmp_path))
    import test_multihalo_gpu as mh
    from neptune_hip import lowering
    shape = (9, 50, 272 if elem == "f64" else 528)       # > one window along J and K in both element types
    acc = [(0, o) for o in mh.star(3, 1)]
    text = mh.module_text(shape, elem, 1, acc, [1, 1, 1], [n - 1 for n in shape])
    mod = lowering.compile_module(text)
    entry = mod.geom_entry("resid")
    assert entry.fn2 is not None and entry.fn3 is not None
    npdt = np.float32 if elem == "f32" else np.float64
    u = (helpers.hash_field(shape, npdt, seed=13) * npdt(0.01)).astype(npdt)
    bounds = ([1, 1, 1], [n - 1 for n in shape])
    m = helpers.oracle.Module.parse(text)
    chain = [u]
    for _ in range(3):
        o = np.zeros(shape, npdt)
        m.call("entry", o, chain[-1])
        chain.append(o)
    fin = nh.fields.DeviceField.from_numpy(u)
    out = nh.fields.DeviceField.empty_like(fin)
    for applies in (2, 3):
        out.tensor.fill_(float("nan"))
        assert nh.apply.apply_twice(entry, fin, out, bounds, applies=applies)
        nh.torch.cuda.synchronize()
        assert helpers.bits_equal(out.numpy(), chain[applies]), f"{applies} applies\n" + helpers.mismatch_report(out.numpy(), chain[applies])


WIDE_CASES = {
    # name: (element, shape, inputs, radii per axis of input 0)
    "radius2_f64": ("f64", (11, 60, 272), 1, (2, 2, 2)),
    "radius2_f32": ("f32", (9, 44, 528), 1, (2, 2, 2)),
    "radii_1_2_1_f64": ("f64", (8, 40, 264), 1, (1, 2, 1)),
    "radii_2_1_2_f64": ("f64", (12, 36, 256), 1, (2, 1, 2)),
    "radius1_with_coefficient_field": ("f64", (9, 50, 272), 2, (1, 1, 1)),
    "radius2_with_two_coefficient_fields_f32": ("f32", (10, 40, 512), 3, (2, 2, 2)),
    # rank 2 (rows marched, no J axis): 9-point stars, unequal radii, coefficient fields; three applies per pass fit at radius 2 too
    "radius2_2d_f64": ("f64", (300, 376), 1, (2, 2)),
    "radii_1_2_2d_f32": ("f32", (200, 528), 1, (1, 2)),
    "radius1_2d_with_coefficient_field": ("f64", (150, 264), 2, (1, 1)),
    "radius2_2d_with_two_coefficient_fields_f32": ("f32", (120, 512), 3, (2, 2)),
}


def wide_case_text(name):
    import test_multihalo_gpu as mh
    elem, shape, nin, rad = WIDE_CASES[name]
    rank = len(shape)
    acc = [(0, (0,) * rank)]
    for d in range(rank):
        for sdist in range(1, rad[d] + 1):
            for sign in (-1, 1):
                o = [0] * rank
                o[d] = sign * sdist
                acc.append((0, tuple(o)))
    acc += [(k, (0,) * rank) for k in range(1, nin)]
    lb, ub = list(rad), [n - r for n, r in zip(shape, rad)]
    return mh.module_text(shape, elem, nin, acc, lb, ub), lb, ub


@pytest.fixture(scope="module")
def wide_cache(tmp_path_factory):
    """the six modules of the wide-chain cases, compiled side by side before the first of them runs"""
    import os
    d = tmp_path_factory.mktemp("neptune_cache_wide_chains")
    saved = os.environ.get("NEPTUNE_CACHE_DIR")
    os.environ["NEPTUNE_CACHE_DIR"] = str(d)
    try:
        helpers.prefetch_modules([wide_case_text(n)[0] for n in WIDE_CASES])
    finally:
        if saved is None:
            os.environ.pop("NEPTUNE_CACHE_DIR", None)
        else:
            os.environ["NEPTUNE_CACHE_DIR"] = saved
    return d


@pytest.mark.parametrize("name", list(WIDE_CASES))
def test_chains_beyond_the_7_point_family(nh, name, wide_cache, monkeypatch):
    """round 3: the chain kernel takes star footprints of input 0 up to radius 2 per axis (13-point 4th-order operators:
    two applies per pass) and further inputs read at the centre only (coefficient fields, the same at every stage):
    out = A(A(u; c); c) in one launch == two launches == the oracle's chained applies, bit for bit; chunk seams, window
    seams along J and K in both element types; a step loop over the pair entry"""
    monkeypatch.setenv("NEPTUNE_CACHE_DIR", str(wide_cache))
    from neptune_hip import lowering
    elem, shape, nin, rad = WIDE_CASES[name]
    text, lb, ub = wide_case_text(name)
    mod = lowering.compile_module(text)
    entry = mod.geom_entry("resid")
    npdt = np.float32 if elem == "f32" else np.float64
    u = (helpers.hash_field(shape, npdt, seed=13) * npdt(0.01)).astype(npdt)
    coef = [(helpers.hash_field(shape, npdt, seed=20 + k) * npdt(0.01)).astype(npdt) for k in range(1, nin)]
    m = helpers.oracle.Module.parse(text)
    chain = [u]
    for _ in range(4):
        o = np.zeros(shape, npdt)
        m.call("entry", o, chain[-1], *coef)
        chain.append(o)
    bounds = (lb, ub)
    fin = nh.fields.DeviceField.from_numpy(u)
    fco = [nh.fields.DeviceField.from_numpy(c) for c in coef]
    out = nh.fields.DeviceField.empty_like(fin)
    import ctypes as C
    g = nh.apply.geom_for([fin] + fco, out, bounds)
    ins = nh.apply._in_array([fin] + fco)
    wide = max(rad) > 1 and len(shape) == 3
    for chunk in (0, 3 if len(shape) == 3 else 40):
        out.tensor.fill_(float("nan"))
        cfg = nh.apply.make_cfg(chunk=chunk)
        assert entry.fn2(C.byref(g), ins, out.ptr, None, C.byref(cfg)) == 0
        nh.torch.cuda.synchronize()
        assert helpers.bits_equal(out.numpy(), chain[2]), f"{name} chunk {chunk}\n" + helpers.mismatch_report(out.numpy(), chain[2])
    # three applies per pass: the radius-1 footprints and every rank-2 one take it, rank-3 radius 2 declines (three rings of
    # five planes do not fit)
    rc3 = entry.fn3(C.byref(g), ins, out.ptr, None, None)
    if wide:
        assert rc3 == nh.capi.EUNSUPPORTED
    else:
        assert rc3 == 0
        nh.torch.cuda.synchronize()
        assert helpers.bits_equal(out.numpy(), chain[3]), helpers.mismatch_report(out.numpy(), chain[3])
    # a step loop: the coefficient fields ride along as the fixed inputs 1..
    with monkeypatch.context() as mp:
        mp.setenv("NEPTUNE_HIP_CHAIN_MIN_CELLS", "0")
        a = nh.fields.DeviceField.from_numpy(u)
        b = nh.fields.DeviceField.empty_like(a)
        res = nh.apply.step_loop(entry, a, b, bounds, 4, others=fco)
        nh.torch.cuda.synchronize()
        assert helpers.bits_equal(res.numpy(), chain[4]), helpers.mismatch_report(res.numpy(), chain[4])


CASES_2D = [
    ((40, 256), None, None),
    ((300, 376), None, None),                          # four column windows, chunk seams along the rows
    ((33, 128), None, None),                           # one wave span
    ((70, 256), [-4, 9], None),
    ((64, 512), None, ([3, 10], [50, 401])),
]


@pytest.mark.parametrize("applies", [2, 3])
@pytest.mark.parametrize("shape,origin,bounds", CASES_2D)
def test_rank2_chains_equal_separate_launches_and_the_oracle(nh, shape, origin, bounds, applies):
    """the 2-D form (waves marching down the rows of a column window, no LDS): 5-point operator"""
    body = nh.capi.BODY_LAP2D5_F64
    lb = origin or [0, 0]
    u = helpers.hash_field(shape, np.float64, seed=29)
    b = ([l + 1 for l in lb], [l + n - 1 for l, n in zip(lb, shape)]) if bounds is None else \
        ([l + x for l, x in zip(lb, bounds[0])], [l + x for l, x in zip(lb, bounds[1])])
    cur = nh.fields.DeviceField.from_numpy(u, lb)
    o = u
    for _ in range(applies):
        nxt = nh.fields.DeviceField.empty_like(cur)
        nh.apply.apply_builtin(body, [cur], nxt, b)
        cur = nxt
        o = helpers.oracle_entry("2d5", o, origin=lb, bounds=b)
    want = cur.numpy()
    assert helpers.bits_equal(want, o)
    fin = nh.fields.DeviceField.from_numpy(u, lb)
    one = nh.fields.DeviceField.empty_like(fin)
    for chunk in (0, 7, 64):
        one.tensor.fill_(float("nan"))
        assert nh.apply.apply_twice(body, fin, one, b, cfg=nh.apply.make_cfg(chunk=chunk), applies=applies)
        nh.torch.cuda.synchronize()
        assert helpers.bits_equal(one.numpy(), want), f"chunk {chunk}\n" + helpers.mismatch_report(one.numpy(), want)


@pytest.mark.parametrize("steps", [3, 5, 8, 31, 58])
def test_rank2_step_loop_and_lowered_euler_step(nh, steps, tmp_path, monkeypatch):
    """2-D explicit time loop: the built-in operator through neptune_hip_step_loop, and a lowered module's fused Euler step
    (time_advance + its rhs apply) through its chain entries, against the oracle stepping one call at a time"""
    monkeypatch.setenv("NEPTUNE_CACHE_DIR", str(tmp_path))
    from neptune_hip import lowering
    shape = (48, 256)
    bounds = ([1, 1], [n - 1 for n in shape])
    u = helpers.hash_field(shape, np.float64, seed=6) * 0.1
    a = nh.fields.DeviceField.from_numpy(u)
    b = nh.fields.DeviceField.empty_like(a)
    res = nh.apply.step_loop(nh.capi.BODY_LAP2D5_F64, a, b, bounds, steps)
    nh.torch.cuda.synchronize()
    o = u
    for _ in range(steps):
        o = helpers.oracle_entry("2d5", o)
    assert res is (b if steps % 2 else a) and helpers.bits_equal(res.numpy(), o)
    text = helpers.stencil_module("2d5", list(shape), time_step=0.0625)
    mod = lowering.compile_module(text)
    entry = mod.geom_entry("step")
    assert entry.fn2 is not None and entry.fn3 is not None
    a = nh.fields.DeviceField.from_numpy(u)
    b = nh.fields.DeviceField.empty_like(a)
    res = nh.apply.step_loop(entry, a, b, bounds, steps)
    nh.torch.cuda.synchronize()
    m = helpers.oracle.Module.parse(text)
    ha, hb = u.copy(), np.zeros(shape)
    for _ in range(steps):
        m.call("step", hb, ha)
        ha, hb = hb, ha
    assert helpers.bits_equal(res.numpy(), ha), helpers.mismatch_report(res.numpy(), ha)

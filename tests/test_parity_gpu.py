"""Parity proper: the HIP path, called through the C ABI, against the oracle on the same inputs.

Bar: BIT-EXACT for every case (integer-like discipline on floating point: the device code is
built -ffp-contract=off and evaluates the body's ops in textual order, exactly like the
reference's FMA-free scalar lowering).  No tolerance is used anywhere in this file.
"""
import ctypes as C

import numpy as np
import pytest

import helpers
from helpers import bits_equal, mismatch_report

pytestmark = pytest.mark.gpu

KIND_BODY = {"2d5": "lap2d5_f64", "3d7": "lap3d7_f64", "3d27": "lap3d27_f32"}
KIND_DTYPE = {"2d5": np.float64, "3d7": np.float64, "3d27": np.float32}


@pytest.fixture(scope="module")
def nh(built_libs):
    import torch
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from neptune_hip import _capi, apply, fields
    lib = _capi.load()
    assert lib.neptune_hip_available() == 1
    lib.neptune_hip_init(0)
    assert lib.neptune_hip_arch().decode().startswith("gfx950"), lib.neptune_hip_arch()

    class NS:
        pass
    ns = NS()
    ns.capi, ns.apply, ns.fields, ns.lib, ns.torch = _capi, apply, fields, lib, torch
    return ns


def _run(nh, kind, u, cfg=None, origin=None, bounds=None, region=None, prefill=None):
    body = nh.apply.BODY_BY_NAME[KIND_BODY[kind]]
    origin = [0] * u.ndim if origin is None else origin
    fin = nh.fields.DeviceField.from_numpy(u, origin)
    fout = nh.fields.DeviceField.empty_like(fin)
    if prefill is not None:
        fout.tensor.fill_(prefill)
    if bounds is None:
        bounds = ([o + 1 for o in origin], [o + n - 1 for o, n in zip(origin, u.shape)])
    nh.apply.apply_builtin(body, [fin], fout, bounds, region=region, cfg=cfg)
    nh.torch.cuda.synchronize()
    return fout.numpy()


def _variants(nh, rank):
    return range(nh.lib.neptune_hip_march_variant_count(rank))


SHAPES = {
    # incl. ragged rows (last extent not a multiple of the 16-byte lane vector: 2^k+1 node-centred grids)
    "2d5": [(3, 3), (5, 8), (17, 130), (40, 256), (33, 1024), (64, 1280), (9, 37), (33, 1025), (19, 131), (7, 5)],
    "3d7": [(3, 3, 4), (8, 8, 8), (6, 7, 130), (20, 18, 128), (33, 17, 256), (12, 70, 384), (5, 9, 11), (9, 10, 257),
            (6, 7, 131), (17, 9, 129),
            # a row or two past the last whole row tile (2^k+1 rows): those rows leave the march launch for a direct one
            (6, 33, 130), (5, 66, 131), (4, 129, 128)],
    "3d27": [(3, 3, 4), (8, 8, 8), (6, 7, 132), (20, 18, 256), (9, 33, 512), (5, 9, 11), (6, 7, 133), (5, 9, 259),
             (7, 6, 130), (4, 5, 9), (5, 65, 260), (5, 34, 133)],
}


def _kernel_cfg(nh, kernel):
    """"direct" = rows form (a workgroup per row chunk, scalar row addressing), "direct-flat" = one lane per
    cell with 64-bit index math (the form that takes fields beyond 2^31 rows), "march" """
    if kernel == "march":
        return nh.apply.make_cfg(nh.capi.KERNEL_MARCH)
    return nh.apply.make_cfg(nh.capi.KERNEL_DIRECT, flags=nh.capi.FLAG_DIRECT_FLAT if kernel == "direct-flat" else 0)


@pytest.mark.parametrize("form", ["direct", "direct-flat"])
@pytest.mark.parametrize("kind", ["2d5", "3d7", "3d27"])
def test_direct_kernel_bit_exact(nh, kind, form):
    for shape in SHAPES[kind]:
        u = helpers.hash_field(shape, KIND_DTYPE[kind], seed=5)
        want = helpers.oracle_entry(kind, u)
        got = _run(nh, kind, u, _kernel_cfg(nh, form))
        assert bits_equal(got, want), f"{kind} {shape} {form}\n" + mismatch_report(got, want)


@pytest.mark.parametrize("kind", ["2d5", "3d7", "3d27"])
def test_march_kernel_every_variant_bit_exact(nh, kind):
    vk = 16 // np.dtype(KIND_DTYPE[kind]).itemsize
    rank = len(SHAPES[kind][0])
    for shape in SHAPES[kind]:
        if shape[-1] < 3 * vk:
            continue  # the march kernel needs at least one whole lane vector to store (ragged rows: two)
        u = helpers.hash_field(shape, KIND_DTYPE[kind], seed=6)
        want = helpers.oracle_entry(kind, u)
        for v in _variants(nh, rank):
            for chunk in (0, 1, 3):
                got = _run(nh, kind, u, nh.apply.make_cfg(nh.capi.KERNEL_MARCH, v, chunk), prefill=7.0)
                name = nh.lib.neptune_hip_march_variant_name(rank, v).decode()
                assert bits_equal(got, want), f"{kind} {shape} march {name} chunk={chunk}\n" + mismatch_report(got, want)


def test_auto_plan_runs_march_on_wide_rows_and_direct_on_narrow(nh):
    u = helpers.hash_field((12, 10, 256), np.float64, seed=2)
    fin = nh.fields.DeviceField.from_numpy(u)
    fout = nh.fields.DeviceField.empty_like(fin)
    b = ([1, 1, 1], [11, 9, 255])
    assert nh.apply.plan_builtin(nh.capi.BODY_LAP3D7_F64, [fin], fout, b) == nh.capi.KERNEL_MARCH
    v = helpers.hash_field((12, 10, 30), np.float64, seed=2)
    fin2 = nh.fields.DeviceField.from_numpy(v)
    fout2 = nh.fields.DeviceField.empty_like(fin2)
    assert nh.apply.plan_builtin(nh.capi.BODY_LAP3D7_F64, [fin2], fout2, ([1, 1, 1], [11, 9, 29])) == nh.capi.KERNEL_DIRECT


@pytest.mark.parametrize("kernel", ["direct", "direct-flat", "march"])
def test_sub_box_bounds_and_shifted_logical_origin(nh, kernel):
    """copy-through outside apply.bounds, logical origin != 0 (DataflowLowering.cpp:367-369,401-404,437-440)"""
    u = helpers.hash_field((14, 12, 136), np.float64, seed=9)
    origin = [5, -3, 7]
    bounds = ([7, -1, 10], [16, 6, 139])  # strictly inside; leaves thick copy-through margins
    want = helpers.oracle_entry("3d7", u, origin, bounds)
    got = _run(nh, "3d7", u, _kernel_cfg(nh, kernel), origin, bounds)
    assert bits_equal(got, want), mismatch_report(got, want)
    # margins are input 0, bit for bit
    assert bits_equal(got[0], u[0]) and bits_equal(got[:, :2, :], u[:, :2, :])
    u2 = helpers.hash_field((21, 264), np.float64, seed=10)
    want2 = helpers.oracle_entry("2d5", u2, [100, -50], ([103, -40], [119, 200]))
    got2 = _run(nh, "2d5", u2, _kernel_cfg(nh, kernel), [100, -50], ([103, -40], [119, 200]))
    assert bits_equal(got2, want2), mismatch_report(got2, want2)
    # empty bounds: pure copy-through
    got3 = _run(nh, "2d5", u2, _kernel_cfg(nh, kernel), [0, 0], ([4, 4], [4, 9]))
    assert bits_equal(got3, u2)


@pytest.mark.parametrize("kernel", ["direct", "direct-flat", "march"])
def test_regions_tile_one_apply_without_touching_other_cells(nh, kernel):
    """the slab decomposition launches edge planes and interior separately: the union must equal
    one whole-field apply and cells outside a region must keep their old value"""
    shape = (18, 10, 128)
    u = helpers.hash_field(shape, np.float64, seed=12)
    want = helpers.oracle_entry("3d7", u)
    body = nh.capi.BODY_LAP3D7_F64
    fin = nh.fields.DeviceField.from_numpy(u)
    fout = nh.fields.DeviceField.empty_like(fin)
    fout.tensor.fill_(-777.0)
    bounds = ([1, 1, 1], [17, 9, 127])
    cfg = _kernel_cfg(nh, kernel)
    full = lambda lo, hi: ([lo, 0, 0], [hi, shape[1], shape[2]])
    nh.apply.apply_builtin(body, [fin], fout, bounds, region=full(0, 2), cfg=cfg)
    nh.torch.cuda.synchronize()
    part = fout.numpy()
    assert bits_equal(part[:2], want[:2]) and np.all(part[2:] == -777.0)
    nh.apply.apply_builtin(body, [fin], fout, bounds, region=full(16, 18), cfg=cfg)
    nh.apply.apply_builtin(body, [fin], fout, bounds, region=full(2, 16), cfg=cfg)
    nh.apply.apply_builtin(body, [fin], fout, bounds, region=full(5, 5), cfg=cfg)   # empty region: no-op
    nh.torch.cuda.synchronize()
    got = fout.numpy()
    assert bits_equal(got, want), mismatch_report(got, want)


def test_reference_smoke_known_answers_on_the_device(nh):
    """KAT-1 / KAT-1b: @ac_lap of the reference's smoke_time_advance.mlir, n = 16, driver input"""
    doc = helpers.load_kats()
    for k in doc["kats"]:
        if k["symbol"] != "kat_lap":
            continue
        u = np.array([float.fromhex(h) for h in k["inputs"][0]])
        want = np.array([float.fromhex(h) for h in k["expected"]])
        fin = nh.fields.DeviceField.from_numpy(u)
        fout = nh.fields.DeviceField.empty_like(fin)
        nh.apply.apply_builtin(nh.capi.BODY_LAP1D3_F64, [fin], fout, ([1], [15]))
        nh.torch.cuda.synchronize()
        assert bits_equal(fout.numpy(), want), k["name"]


def test_plan_errors_surface_before_any_launch(nh):
    u = np.arange(1, 5, dtype=np.float64)
    fin = nh.fields.DeviceField.from_numpy(u)
    fout = nh.fields.DeviceField.empty_like(fin)
    # reference smoke_apply.mlir: offsets +-1 over the full range -> out of bounds
    with pytest.raises(nh.capi.NeptuneHipError, match="EOOB"):
        nh.apply.apply_builtin(nh.capi.BODY_LAP1D3_F64, [fin], fout, ([0], [4]))
    # result aliasing an input (the reference always materialises a fresh buffer)
    with pytest.raises(nh.capi.NeptuneHipError, match="EINVAL"):
        nh.apply.apply_builtin(nh.capi.BODY_LAP1D3_F64, [fin], fin, ([1], [3]))


def test_store_full_and_box(nh):
    s = helpers.hash_field((6, 5), np.float64, seed=3)
    fs = nh.fields.DeviceField.from_numpy(s, (2, 0))
    fd = nh.fields.DeviceField((0, -1), (10, 6), nh.capi.F64)
    fd.tensor.zero_()
    nh.apply.store(fs, fd, ([3, 1], [7, 4]))
    nh.torch.cuda.synchronize()
    want = np.zeros((10, 7))
    want[3:7, 2:5] = s[1:5, 1:4]
    assert bits_equal(fd.numpy(), want)
    f2 = nh.fields.DeviceField.empty_like(fs)
    nh.apply.store(fs, f2)
    nh.torch.cuda.synchronize()
    assert bits_equal(f2.numpy(), s)
    # whole-buffer stores of every length around the 16-byte words of the streaming copy kernel (tail bytes, misaligned
    # buffers: the runtime's copy), the destination's neighbours untouched
    torch = nh.torch
    for dt, code in ((torch.float32, nh.capi.F32), (torch.float64, nh.capi.F64)):
        for n in (1, 2, 3, 4, 5, 7, 255, 1024, 1027, 4099, 70001):
            for shift in (0, 1):
                src = torch.arange(n + 8, dtype=dt, device="cuda") * 0.5 + 1.0
                dst = torch.full((n + 8,), -1.0, dtype=dt, device="cuda")
                a, b = src[shift:shift + n], dst[4 + shift:4 + shift + n]
                assert nh.lib.neptune_hip_store_full(code, a.data_ptr(), b.data_ptr(), n, None) == 0
                torch.cuda.synchronize()
                want_d = torch.full((n + 8,), -1.0, dtype=dt, device="cuda")
                want_d[4 + shift:4 + shift + n] = a
                assert torch.equal(dst, want_d), (dt, n, shift)
    # 3-D f32 box, empty box, and a box leaving the destination
    s3 = helpers.hash_field((4, 6, 10), np.float32, seed=4)
    f3 = nh.fields.DeviceField.from_numpy(s3)
    d3 = nh.fields.DeviceField((-1, -1, -1), (5, 7, 11), nh.capi.F32)
    d3.tensor.fill_(9.0)
    nh.apply.store(f3, d3, ([1, 2, 3], [3, 5, 9]))
    nh.apply.store(f3, d3, ([1, 2, 3], [1, 5, 9]))
    nh.torch.cuda.synchronize()
    want3 = np.full((6, 8, 12), 9.0, np.float32)
    want3[2:4, 3:6, 4:10] = s3[1:3, 2:5, 3:9]
    assert bits_equal(d3.numpy(), want3)
    with pytest.raises(nh.capi.NeptuneHipError, match="EOOB"):
        nh.apply.store(f3, d3, ([0, 0, 0], [4, 6, 11]))


def test_device_fill_matches_host_twin_and_mismatch_counter(nh):
    for dt, code in ((np.float64, nh.capi.F64), (np.float32, nh.capi.F32)):
        f = nh.fields.DeviceField.hashed((7, 9, 33), code, seed=77, index_offset=1000)
        nh.torch.cuda.synchronize()
        assert bits_equal(f.numpy(), helpers.hash_field((7, 9, 33), dt, 77, 1000))
        g = nh.fields.DeviceField.hashed((7, 9, 33), code, seed=77, index_offset=1000)
        assert nh.apply.count_mismatch(f, g) == 0
        g.tensor.view(-1)[5] += 1
        g.tensor.view(-1)[500] = float("nan")
        assert nh.apply.count_mismatch(f, g) == 2


# ---------------------------------------------------------------------------------------------
# BASELINE.json's full sizes: too big for the CPU oracle as a whole, so parity is checked through
#   (a) march kernel == direct kernel on every cell (two independent device implementations),
#   (b) oracle parity on sampled planes / row bands (3 input planes suffice for one output plane),
#   (c) copy-through faces equal the input bit for bit.
# ---------------------------------------------------------------------------------------------
def _check_planes_3d(nh, kind, fin, fout, planes):
    n0 = fin.shape[0]
    for i in planes:
        if i == 0 or i == n0 - 1:
            assert bits_equal(fout.planes(i, i + 1), fin.planes(i, i + 1)), f"plane {i}: copy-through"
            continue
        slab = fin.planes(i - 1, i + 2)
        want = helpers.oracle_entry(kind, slab)[1]
        got = fout.planes(i, i + 1)[0]
        assert bits_equal(got, want), f"plane {i}\n" + mismatch_report(got, want)


@pytest.mark.parametrize("kind,shape,code", [("3d7", (1024, 1024, 1024), 0), ("3d7", (512, 512, 512), 0),
                                             ("3d27", (512, 512, 512), 1)])
def test_full_size_3d(nh, kind, shape, code):
    body = nh.apply.BODY_BY_NAME[KIND_BODY[kind]]
    fin = nh.fields.DeviceField.hashed(shape, code, seed=2024)
    f_m = nh.fields.DeviceField.empty_like(fin)
    f_d = nh.fields.DeviceField.empty_like(fin)
    bounds = ([1, 1, 1], [n - 1 for n in shape])
    assert nh.apply.plan_builtin(body, [fin], f_m, bounds) == nh.capi.KERNEL_MARCH
    nh.apply.apply_builtin(body, [fin], f_m, bounds)
    nh.apply.apply_builtin(body, [fin], f_d, bounds, cfg=nh.apply.make_cfg(nh.capi.KERNEL_DIRECT))
    assert nh.apply.count_mismatch(f_m, f_d) == 0
    n0 = shape[0]
    _check_planes_3d(nh, kind, fin, f_m, [0, 1, 2, n0 // 2 - 1, n0 // 2, n0 - 2, n0 - 1, 127, 128, 129])
    del f_d, f_m, fin
    nh.torch.cuda.empty_cache()


def test_full_size_2d_8192(nh):
    shape = (8192, 8192)
    fin = nh.fields.DeviceField.hashed(shape, nh.capi.F64, seed=8192)
    f_m = nh.fields.DeviceField.empty_like(fin)
    f_d = nh.fields.DeviceField.empty_like(fin)
    bounds = ([1, 1], [8191, 8191])
    body = nh.capi.BODY_LAP2D5_F64
    assert nh.apply.plan_builtin(body, [fin], f_m, bounds) == nh.capi.KERNEL_MARCH
    nh.apply.apply_builtin(body, [fin], f_m, bounds)
    nh.apply.apply_builtin(body, [fin], f_d, bounds, cfg=nh.apply.make_cfg(nh.capi.KERNEL_DIRECT))
    assert nh.apply.count_mismatch(f_m, f_d) == 0
    # the whole field fits the numpy oracle (512 MiB): full parity
    got = f_m.numpy()
    want = helpers.oracle_entry("2d5", fin.numpy())
    assert bits_equal(got, want), mismatch_report(got, want)


@pytest.mark.parametrize("kind,shape", [("3d7", (1024, 1024, 1024)), ("3d7", (1025, 513, 1025)), ("3d27", (512, 512, 512)), ("2d5", (8192, 8192)),
                                        ("2d5", (8193, 4097))])
def test_full_size_polynomial_fields_every_cell(nh, kind, shape):
    """size-independent properties checked on EVERY cell of the full-size fields: a Laplacian-type stencil of an
    affine field with integer coefficients is exactly 0 at every interior cell (all partial sums are integers far
    below 2^53 / 2^24), of a quadratic field exactly its constant second difference, and the rim is the input --
    any wrong neighbour, plane, halo row, tile seam, chunk seam or tail cell shows up as a nonzero"""
    torch = nh.torch
    body = nh.apply.BODY_BY_NAME[KIND_BODY[kind]]
    tdt = torch.float64 if KIND_DTYPE[kind] == np.float64 else torch.float32
    rank = len(shape)
    coef = [3, 5, 7][:rank]                                    # affine: u = 3 i + 5 j + 7 k  (< 2^15: exact in f32 too)
    idx = [torch.arange(n, device="cuda", dtype=tdt) for n in shape]
    bounds = ([1] * rank, [n - 1 for n in shape])
    inner = tuple(slice(1, -1) for _ in shape)
    for quadratic in (False, True):
        if quadratic and tdt == torch.float32:
            continue                                           # i^2 terms leave the exact f32 integer range at 512
        u = torch.zeros(shape, dtype=tdt, device="cuda")
        for d in range(rank):
            view = [1] * rank
            view[d] = shape[d]
            x = idx[d].reshape(view)
            u += (x * x * (d + 1)) if quadratic else (x * coef[d])
        fin = nh.fields.DeviceField(tuple([0] * rank), tuple(shape), nh.capi.F64 if tdt == torch.float64 else nh.capi.F32, u)
        fout = nh.fields.DeviceField.empty_like(fin)
        fout.tensor.fill_(-1.0)
        assert nh.apply.plan_builtin(body, [fin], fout, bounds) == nh.capi.KERNEL_MARCH
        nh.apply.apply_builtin(body, [fin], fout, bounds)
        torch.cuda.synchronize()
        out = fout.tensor
        # second differences of d*(x^2) summed over dims = 2*(1+2+3), times the fixture's dxinv2 (a power of two)
        scale = {"2d5": 0.125, "3d7": 0.0625, "3d27": 0.015625}[kind]
        expect = scale * 2.0 * sum(range(1, rank + 1)) if quadratic else 0.0
        bad = int((out[inner] != expect).sum())
        assert bad == 0, f"{kind} {shape} quadratic={quadratic}: {bad} interior cells differ from {expect}"
        rim = torch.ones(shape, dtype=torch.bool, device="cuda")
        rim[inner] = False
        assert bool((out[rim] == u[rim]).all()), "copy-through rim differs from input 0"
        del u, fin, fout, out, rim
        torch.cuda.empty_cache()


@pytest.mark.parametrize("form", ["direct", "direct-flat"])
def test_direct_kernel_beyond_2_pow_32_work_items(nh, form):
    """one workgroup per 256 cells over 1.2e9 cells is more than 2^32 work-items, which HIP refuses in one grid
    dimension ("invalid configuration argument", found on a 2048^3 field): such launches fold the workgroup count
    into (x, y).  Affine integer field -> exactly 0 on every interior cell."""
    torch = nh.torch
    shape = (1100, 1000, 1100)
    u = torch.zeros(shape, dtype=torch.float64, device="cuda")
    for d, c in enumerate((3, 5, 7)):
        view = [1, 1, 1]
        view[d] = shape[d]
        u += torch.arange(shape[d], device="cuda", dtype=torch.float64).reshape(view) * c
    fin = nh.fields.DeviceField((0, 0, 0), shape, nh.capi.F64, u)
    fout = nh.fields.DeviceField.empty_like(fin)
    fout.tensor.fill_(-1.0)
    nh.apply.apply_builtin(nh.capi.BODY_LAP3D7_F64, [fin], fout, ([1, 1, 1], [n - 1 for n in shape]), cfg=_kernel_cfg(nh, form))
    torch.cuda.synchronize()
    out = fout.tensor
    assert int((out[1:-1, 1:-1, 1:-1] != 0).sum()) == 0
    assert bool((out[0] == u[0]).all() and (out[-1] == u[-1]).all() and (out[:, :, -1] == u[:, :, -1]).all())
    del u, fin, fout, out
    torch.cuda.empty_cache()


def test_config1_1024x1024_fixture_geometry(nh):
    """BASELINE.json configs[0]: apply-2d-5pt.mlir, 1024x1024 f64 (the CPU-runnable case)"""
    u = helpers.hash_field((1024, 1024), np.float64, seed=1)
    want = helpers.oracle_entry("2d5", u)
    got = _run(nh, "2d5", u)
    assert bits_equal(got, want), mismatch_report(got, want)
    assert bits_equal(helpers.c_oracle_entry("2d5", u), want)


def test_randomised_geometries_bit_exact(nh):
    """seeded sweep over shapes, logical origins and apply bounds (including bounds that touch the
    faces where the stencil still fits, thin slabs, partial waves, rows narrower than a wave):
    both kernels against the oracle on every case"""
    rng = np.random.default_rng(20261004)
    cases = 0
    for kind in ("2d5", "3d7", "3d27"):
        rank = 2 if kind == "2d5" else 3
        dt = KIND_DTYPE[kind]
        vk = 16 // np.dtype(dt).itemsize
        for _ in range(14):
            shape = [int(rng.integers(3, 24)) for _ in range(rank - 1)]
            last = int(rng.choice([4, 6, 8, 30, 64, 126, 128, 130, 200, 256, 300, 384]))
            if rng.integers(0, 2):
                last = last // vk * vk                      # aligned rows
            else:
                last += int(rng.integers(1, vk))            # ragged rows: unaligned vector accesses + tail launch
            shape.append(max(last, vk * 3))
            origin = [int(rng.integers(-5, 6)) for _ in range(rank)]
            lb, ub = [], []
            for d in range(rank):
                lo = int(rng.integers(1, max(2, shape[d] // 2)))
                hi = int(rng.integers(lo, shape[d]))          # may be empty (lo == hi)
                lb.append(origin[d] + lo)
                ub.append(origin[d] + min(hi, shape[d] - 1))
                if ub[-1] < lb[-1]:
                    ub[-1] = lb[-1]
            u = helpers.hash_field(tuple(shape), dt, seed=int(rng.integers(1, 1 << 30)))
            want = helpers.oracle_entry(kind, u, origin, (lb, ub))
            for kern in ("direct", "direct-flat", "march"):
                got = _run(nh, kind, u, _kernel_cfg(nh, kern), origin, (lb, ub), prefill=3.0)
                assert bits_equal(got, want), f"{kind} shape={shape} origin={origin} bounds={(lb, ub)} kernel={kern}\n" + \
                    mismatch_report(got, want)
            cases += 1
    assert cases == 42


LAP1D = '''
#l = #neptune_ir.location<"cell">
!t = !neptune_ir.temp<element = f64, bounds = #neptune_ir.bounds<lb = [{lo}], ub = [{hi}]>, location = #l>
module {{
  neptune_ir.linear_opdef @ac_lap : (!t) -> !t {{
  ^bb0(%u: !t):
    %r = neptune_ir.apply(%u) attributes {{bounds = #neptune_ir.bounds<lb = [{blo}], ub = [{bhi}]>}} : (!t) -> !t {{
      ^bb0(%i: index, %a: !t):
        %um1 = neptune_ir.access %a[-1] : !t -> f64
        %u0 = neptune_ir.access %a[0] : !t -> f64
        %up1 = neptune_ir.access %a[1] : !t -> f64
        %two = arith.constant 2.0 : f64
        %dxinv2 = arith.constant 100.0 : f64
        %t0 = arith.mulf %two, %u0 : f64
        %t1 = arith.subf %um1, %t0 : f64
        %t2 = arith.addf %t1, %up1 : f64
        %lap_i = arith.mulf %dxinv2, %t2 : f64
        neptune_ir.yield %lap_i : f64
    }}
    neptune_ir.return %r : !t
  }}
}}
'''


@pytest.mark.parametrize("n,origin,bounds", [(16, 0, (1, 15)), (130, -7, (-6, 122)), (4096, 0, (1, 4095)), (100000, 3, (500, 99000)),
                                             (4097, 0, (1, 4096)), (100001, -2, (-1, 99998))])
def test_rank1_fields_on_both_kernels(nh, n, origin, bounds):
    """the reference's own inputs are 1-D (@ac_lap, smoke_time_advance.mlir:13-29): a long 1-D field is one
    row, every wave takes 1 KiB of it (march kernel, single step), neighbours by wave shifts"""
    text = LAP1D.format(lo=origin, hi=origin + n, blo=bounds[0], bhi=bounds[1])
    u = helpers.hash_field((n,), np.float64, seed=n)
    want = helpers.oracle.Module.parse(text).call("ac_lap", u)
    fin = nh.fields.DeviceField.from_numpy(u, (origin,))
    for kern in (nh.capi.KERNEL_DIRECT, nh.capi.KERNEL_MARCH, nh.capi.KERNEL_AUTO):
        fout = nh.fields.DeviceField.empty_like(fin)
        fout.tensor.fill_(9.0)
        cfg = nh.apply.make_cfg(kern)
        nh.apply.apply_builtin(nh.capi.BODY_LAP1D3_F64, [fin], fout, ([bounds[0]], [bounds[1]]), cfg=cfg)
        nh.torch.cuda.synchronize()
        assert bits_equal(fout.numpy(), want), f"n={n} kernel={kern}\n" + mismatch_report(fout.numpy(), want)
    want_plan = nh.capi.KERNEL_MARCH if n >= 128 else nh.capi.KERNEL_DIRECT
    assert nh.apply.plan_builtin(nh.capi.BODY_LAP1D3_F64, [fin], fout, ([bounds[0]], [bounds[1]])) == want_plan


@pytest.mark.parametrize("kind,shape,steps", [("2d5", (64, 256), 37), ("3d7", (10, 9, 128), 12), ("2d5", (33, 130), 3), ("3d7", (8, 8, 8), 101)])
def test_graph_captured_step_loop_equals_chained_launches(nh, kind, shape, steps):
    """neptune_hip_step_loop: the ping-pong pair captured once into a hipGraph and replayed == `steps` plain launches
    == the oracle's chained applies; a second call reuses the cached graph; odd/even step counts end in the right field"""
    body = nh.apply.BODY_BY_NAME[KIND_BODY[kind]]
    u = helpers.hash_field(shape, KIND_DTYPE[kind], seed=41) * 0.25          # keep 100 Laplacian steps finite
    want = u
    for _ in range(steps):
        want = helpers.oracle_entry(kind, want)
    bounds = ([1] * len(shape), [n - 1 for n in shape])
    for rep in range(2):
        a = nh.fields.DeviceField.from_numpy(u) if rep == 0 else a0
        a0 = a
        if rep == 1:
            a.tensor.copy_(nh.torch.from_numpy(u))
        b = nh.fields.DeviceField.empty_like(a) if rep == 0 else b0
        b0 = b
        last = nh.apply.step_loop(body, a, b, bounds, steps)
        nh.torch.cuda.synchronize()
        assert last is (b if steps % 2 else a)
        assert bits_equal(last.numpy(), want), f"rep {rep}\n" + mismatch_report(last.numpy(), want)


@pytest.mark.parametrize("kind", ["2d5", "3d7", "3d27"])
def test_special_values_denormals_signed_zeros_infinities_nans(nh, kind):
    """IEEE corner values through every kernel form: subnormal inputs and results are kept (nothing flushes to zero, fp32
    included), signed zeros and infinities come out bit for bit like the CPU restatement; where the oracle has a NaN the GPU
    has a NaN (payloads of generated NaNs are the hardware's business on both sides)"""
    dt = KIND_DTYPE[kind]
    shape = {"2d5": (40, 520), "3d7": (10, 18, 260), "3d27": (9, 12, 520)}[kind]
    rng = np.random.default_rng(11)
    u = helpers.hash_field(shape, dt, seed=8)
    tiny = np.finfo(dt).tiny
    specials = np.array([tiny / 8, -tiny / 3, tiny * 1.5, np.nextafter(dt(0), dt(1)), 0.0, -0.0, np.inf, -np.inf, np.nan,
                         np.finfo(dt).max, -np.finfo(dt).max / 2, tiny], dtype=dt)
    flat = u.reshape(-1)
    where = rng.choice(flat.size, size=flat.size // 7, replace=False)
    flat[where] = specials[rng.integers(0, len(specials), size=where.size)]
    # a block of nothing but subnormals: results there are subnormal sums, not zeros
    sl = tuple(slice(2, 6) for _ in shape[:-1]) + (slice(64, 200),)
    u[sl] = (rng.integers(1, 1000, size=u[sl].shape) * np.nextafter(dt(0), dt(1))).astype(dt)
    with np.errstate(all="ignore"):
        want = helpers.oracle_entry(kind, u)
    assert np.any((np.abs(want[sl]) > 0) & (np.abs(want[sl]) < tiny)), "the fixture should produce subnormal results"
    rank = len(shape)
    cfgs = [None, _kernel_cfg(nh, "direct"), _kernel_cfg(nh, "direct-flat")] + \
           [nh.apply.make_cfg(nh.capi.KERNEL_MARCH, v, c) for v in range(min(8, nh.lib.neptune_hip_march_variant_count(rank))) for c in (0, 3)]
    ui = np.dtype(dt).itemsize
    as_int = np.uint64 if ui == 8 else np.uint32
    for cfg in cfgs:
        got = _run(nh, kind, u, cfg, prefill=5.0)
        nan_w, nan_g = np.isnan(want), np.isnan(got)
        assert np.array_equal(nan_w, nan_g), f"{kind} {cfg and (cfg.kernel, cfg.variant, cfg.chunk)}: NaN positions differ"
        same = got.view(as_int)[~nan_w] == want.view(as_int)[~nan_w]
        assert same.all(), f"{kind} {cfg and (cfg.kernel, cfg.variant, cfg.chunk)}: {np.count_nonzero(~same)} finite / infinite / zero cells differ in bits"

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


def test_reduce_rejects_bad_domains(nh):
    f = nh.fields.DeviceField.from_numpy(np.ones((4, 4)))
    with pytest.raises(nh.capi.NeptuneHipError, match="EOOB"):
        nh.apply.reduce_sum(f, ((0, 0), (5, 4)))

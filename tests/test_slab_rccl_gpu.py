"""include/neptune_hip.h section 8 on ONE GPU: the C-ABI halo exchange and sharded apply executed for real, on both
transports -- a communicator of world size 1 whose neighbours are the rank itself (loop-back), so ncclSend / ncclRecv
(or the peer transport's handshake kernels and pushes), the communication stream, the events and the
exchange-beside-interior schedule all run.  RCCL matches the two send/receive pairs in order, so its loop-back fills
the lower ghost planes with the rank's own FIRST owned planes and the upper ones with its LAST; the peer transport
pushes to the neighbour's opposite side, a periodic wrap: lower ghosts = LAST owned planes.  The oracle is run on
exactly the array each implies.  (Two ranks cannot share one GPU under RCCL; they can on the peer transport:
tests/test_slab_peer_gpu.py runs the C path with two and three processes.)"""
import ctypes as C

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cache_dir(tmp_path_factory):
    return tmp_path_factory.mktemp("neptune_cache_slab_c")   # modules compile once for both transports


def loopback(local, lo, hi, r, transport):
    """the local buffer after a loop-back exchange of its r ghost planes per side"""
    ext = local.copy()
    if transport == "rccl":
        ext[:lo], ext[hi:] = local[lo:lo + r], local[hi - r:hi]
    else:
        ext[:lo], ext[hi:] = local[hi - r:hi], local[lo:lo + r]
    return ext


@pytest.fixture(scope="module", params=["rccl", "peer"])
def nh(request):
    import torch
    from neptune_hip import _capi, apply, fields, slab

    class NS:
        pass
    ns = NS()
    ns.torch, ns.capi, ns.apply, ns.fields, ns.slab = torch, _capi, apply, fields, slab
    ns.lib = _capi.load()
    ns.lib.neptune_hip_init(0)
    torch.cuda.set_device(0)
    ns.transport = request.param
    ns.comm = slab.SlabComm(0, 1, transport=request.param)          # world of one rank: no unique id needed
    assert ns.lib.neptune_hip_slab_comm_transport(ns.comm.ptr).decode() == request.param
    yield ns
    ns.comm.status()                       # no device-side wait of the peer transport ever timed out
    ns.comm.close()


def _middle_slab(nh, shape_own, radius):
    """a slab that believes it has a neighbour on both sides (ghost planes below and above)"""
    n0 = shape_own[0]
    glb = (0,) + (0,) * (len(shape_own) - 1)
    gub = (n0 + 2 * radius + 20,) + tuple(shape_own[1:])   # global box with room on both sides
    return nh.slab.Slab(rank=0, world=1, radius=radius, glb=glb, gub=gub, start=radius + 3, stop=radius + 3 + n0,
                        r_lo=radius, r_hi=radius)


@pytest.mark.parametrize("radius", [1, 2])
def test_halo_exchange_loopback_fills_the_ghost_planes(nh, radius):
    sl = _middle_slab(nh, (10, 6, 128), radius)
    u = helpers.hash_field(sl.local_shape, np.float64, seed=31)
    lo, hi = sl.owned_planes()
    u[:lo] = np.nan
    u[hi:] = np.nan
    t = nh.torch.from_numpy(u.copy()).cuda()
    nh.comm.exchange(sl, t, peer_lo=0, peer_hi=0)
    nh.torch.cuda.synchronize()
    got = t.cpu().numpy()
    want = loopback(u, lo, hi, radius, nh.transport)
    assert helpers.bits_equal(got, want), helpers.mismatch_report(got, want)
    # several buffers in one grouped exchange / one handshake
    ts = [nh.torch.from_numpy(helpers.hash_field(sl.local_shape, np.float64, seed=32 + k)).cuda() for k in range(3)]
    before = [t.cpu().numpy() for t in ts]
    ptrs = (C.c_void_p * 3)(*[t.data_ptr() for t in ts])
    pb = (C.c_size_t * 3)(*[ts[0][0].numel() * 8] * 3)
    assert nh.lib.neptune_hip_halo_exchange_many(nh.comm.ptr, ptrs, pb, 3, sl.n_own, radius, radius, 0, 0, None) == 0
    nh.torch.cuda.synchronize()
    for t, b in zip(ts, before):
        assert helpers.bits_equal(t.cpu().numpy(), loopback(b, lo, hi, radius, nh.transport))


@pytest.mark.parametrize("side", ["lo", "hi"])
def test_one_sided_loopback_of_an_edge_rank(nh, side):
    """an edge rank of the decomposition has ghost planes on one side only; as a loop-back (bench.py --emulate-rank 0/W) both
    transports return the rank's own edge planes of that side into its ghost planes"""
    r = 1
    own = 10
    glb, gub = (0, 0, 0), (own + 30, 6, 128)
    sl = (nh.slab.Slab(rank=0, world=1, radius=r, glb=glb, gub=gub, start=0, stop=own, r_lo=0, r_hi=r) if side == "hi" else
          nh.slab.Slab(rank=0, world=1, radius=r, glb=glb, gub=gub, start=20, stop=20 + own, r_lo=r, r_hi=0))
    u = helpers.hash_field(sl.local_shape, np.float64, seed=37)
    lo, hi = sl.owned_planes()
    u[:lo] = np.nan
    u[hi:] = np.nan
    t = nh.torch.from_numpy(u.copy()).cuda()
    for _ in range(3):                        # the handshake counters advance on one side only
        nh.comm.exchange(sl, t, peer_lo=0, peer_hi=0)
    nh.torch.cuda.synchronize()
    want = u.copy()
    if side == "hi":
        want[hi:] = u[hi - r:hi]
    else:
        want[:lo] = u[lo:lo + r]
    assert helpers.bits_equal(t.cpu().numpy(), want), helpers.mismatch_report(t.cpu().numpy(), want)


def test_halo_exchange_rejects_bad_requests(nh):
    t = nh.torch.zeros((4, 2, 64), dtype=nh.torch.float64, device="cuda")
    lib = nh.lib
    # a neighbour outside the communicator, a halo deeper than the slab, a null buffer
    assert lib.neptune_hip_halo_exchange(nh.comm.ptr, t.data_ptr(), 1024, 2, 1, 1, 0, 5, None) == nh.capi.EINVAL
    assert lib.neptune_hip_halo_exchange(nh.comm.ptr, t.data_ptr(), 1024, 1, 2, 0, 0, 0, None) == nh.capi.EINVAL
    assert lib.neptune_hip_halo_exchange(nh.comm.ptr, None, 1024, 2, 1, 1, 0, 0, None) == nh.capi.EINVAL
    # nothing to exchange is not an error
    assert lib.neptune_hip_halo_exchange(nh.comm.ptr, t.data_ptr(), 1024, 4, 0, 0, -1, -1, None) == nh.capi.OK


@pytest.mark.parametrize("kind,shape_own,overlap", [("3d7", (12, 10, 256), True), ("3d7", (12, 10, 256), False),
                                                    ("3d27", (9, 12, 256), True), ("2d5", (40, 384), True),
                                                    ("3d7", (2, 8, 128), True)])   # last: thinner than two halos
def test_sharded_apply_through_the_c_plan_matches_the_oracle(nh, kind, shape_own, overlap):
    body = {"3d7": nh.capi.BODY_LAP3D7_F64, "2d5": nh.capi.BODY_LAP2D5_F64, "3d27": nh.capi.BODY_LAP3D27_F32}[kind]
    dtype = nh.apply.BODY_DTYPE[body]
    npdt = np.float32 if dtype == nh.capi.F32 else np.float64
    sl = _middle_slab(nh, shape_own, 1)
    lo, hi = sl.owned_planes()
    local = helpers.hash_field(sl.local_shape, npdt, seed=77)
    local[:lo] = np.nan                     # ghosts poisoned: only the exchange can make them valid
    local[hi:] = np.nan
    fin = nh.fields.DeviceField.from_numpy(local, sl.local_lb)
    fout = nh.fields.DeviceField(sl.local_lb, sl.local_ub, dtype)
    fout.tensor.fill_(float("nan"))
    gbounds = ([1] * len(shape_own), [n - 1 for n in sl.gub])
    op = nh.slab.ShardedApply(sl, body, gbounds, overlap=overlap, comm=nh.comm, peers=(0, 0))
    for _ in range(2):                       # the second call reuses plan, stream and events
        op(fin, fout)
    nh.torch.cuda.synchronize()
    got = fout.numpy()[lo:hi]
    # the oracle on the array the loop-back exchange produces, with the slab's logical origin
    ext = loopback(local, lo, hi, 1, nh.transport)
    lb, ub = sl.clip_bounds(gbounds)
    want = helpers.oracle_entry(kind, ext, origin=list(sl.local_lb), bounds=(lb, ub))[lo:hi]
    assert helpers.bits_equal(got, want), helpers.mismatch_report(got, want)
    # the input's ghost planes now hold the exchanged data, the owned planes are untouched
    assert helpers.bits_equal(fin.numpy(), ext)
    # where a step's time goes (neptune_hip_slab_plan_timing): five timed events per step, read back afterwards
    op.timing(True)
    for _ in range(3):
        op(fin, fout)
    t = op.read_timing()
    assert t is not None and t["steps"] == 3 and t["step_ms"] > 0 and t["exchange_ms"] > 0 and t["edge_wait_ms"] >= 0
    assert t["interior_ms"] <= t["step_ms"] * 1.01
    op.timing(False)
    assert helpers.bits_equal(fout.numpy()[lo:hi], want)


def test_sharded_apply_of_a_lowered_module_through_the_c_plan(nh, cache_dir, monkeypatch):
    """a user stencil (the committed 13-point fixture: radius 2 along dim 0, two ghost planes per side) through its
    geometry-level entry"""
    monkeypatch.setenv("NEPTUNE_CACHE_DIR", str(cache_dir))
    from neptune_hip import lowering
    text = (helpers.FIXTURE_DIR / "apply-3d-13pt.mlir").read_text()
    gshape = (20, 18, 256)
    mod = lowering.compile_module(text)
    entry = mod.geom_entry("lap13")
    assert entry.halo0 == 2
    sl = nh.slab.Slab(rank=0, world=1, radius=2, glb=(0, 0, 0), gub=gshape, start=5, stop=14, r_lo=2, r_hi=2)
    lo, hi = sl.owned_planes()
    local = helpers.hash_field(sl.local_shape, np.float64, seed=5)
    local[:lo] = np.nan
    local[hi:] = np.nan
    fin = nh.fields.DeviceField.from_numpy(local, sl.local_lb)
    fout = nh.fields.DeviceField(sl.local_lb, sl.local_ub, nh.capi.F64)
    gbounds = ([2, 2, 2], [18, 16, 254])
    op = nh.slab.ShardedApply(sl, entry, gbounds, comm=nh.comm, peers=(0, 0))
    op(fin, fout)
    nh.torch.cuda.synchronize()
    # oracle: the global field whose planes [3, 16) are the local buffer after the loop-back exchange
    ext = loopback(local, lo, hi, 2, nh.transport)
    glob = np.zeros(gshape)
    glob[sl.local_lb[0]:sl.local_ub[0]] = ext
    want = helpers.oracle.Module.parse(text).call("lap13", glob)[sl.start:sl.stop]
    got = fout.numpy()[lo:hi]
    assert helpers.bits_equal(got, want), helpers.mismatch_report(got, want)


def test_plan_without_neighbours_is_a_plain_apply(nh):
    """world 1 / no ghost planes: no communicator, no stream, one launch"""
    shape = (8, 8, 128)
    sl = nh.slab.decompose(([0, 0, 0], list(shape)), 1, 0, 1)
    u = helpers.hash_field(shape, np.float64, seed=3)
    fin = nh.fields.DeviceField.from_numpy(u)
    fout = nh.fields.DeviceField.empty_like(fin)
    g = nh.apply.geom_for([fin], fout, ([1, 1, 1], [7, 7, 127]))
    plan = nh.lib.neptune_hip_slab_plan_create(None, None, nh.capi.BODY_LAP3D7_F64, nh.capi.F64, C.byref(g), 1, 0, 0, -1, -1, None)
    assert plan
    ins = nh.apply._in_array([fin])
    assert nh.lib.neptune_hip_slab_apply(plan, ins, fout.ptr, None, 1) == 0
    nh.torch.cuda.synchronize()
    nh.lib.neptune_hip_slab_plan_destroy(plan)
    assert helpers.bits_equal(fout.numpy(), helpers.oracle_entry("3d7", u))


@pytest.mark.parametrize("symbol,overlap", [("entry", True), ("step", True), ("entry", False)])
def test_sharded_module_exchange_beside_the_interior(nh, cache_dir, monkeypatch, symbol, overlap):
    """whole lowered functions on a slab with the exchange left in flight (neptune_hip_set_slab_pending): the
    function's stencil apply does its interior, waits for the halo event, then the planes next to the ghosts.  Loop-back
    RCCL on one GPU; the oracle runs the same function on the global field that the exchange implies."""
    monkeypatch.setenv("NEPTUNE_CACHE_DIR", str(cache_dir))
    from neptune_hip import lowering
    gshape = (26, 10, 256)
    text = helpers.stencil_module("3d7", list(gshape), time_step=0.125)
    mod = lowering.compile_module(text)
    sl = nh.slab.Slab(rank=0, world=1, radius=1, glb=(0, 0, 0), gub=gshape, start=6, stop=18, r_lo=1, r_hi=1)
    lo, hi = sl.owned_planes()
    local = helpers.hash_field(sl.local_shape, np.float64, seed=41)
    local[:lo] = np.nan
    local[hi:] = np.nan
    la = nh.torch.from_numpy(local.copy()).cuda()
    lb = nh.torch.full(sl.local_shape, float("nan"), dtype=nh.torch.float64, device="cuda")
    sm = nh.slab.ShardedModule(mod, sl, comm=nh.comm, overlap=overlap, peers=(0, 0))
    assert sm.call(symbol, lb, la) is lb
    nh.torch.cuda.synchronize()
    assert nh.lib.neptune_hip_get_slab_pending() is None            # consumed by the call, slab view cleared
    ext = loopback(local, lo, hi, 1, nh.transport)
    glob = np.zeros(gshape)
    glob[sl.local_lb[0]:sl.local_ub[0]] = ext
    out = np.zeros(gshape)
    helpers.oracle.Module.parse(text).call(symbol, out, glob)
    got = lb.cpu().numpy()[lo:hi]
    want = out[sl.start:sl.stop]
    assert helpers.bits_equal(got, want), helpers.mismatch_report(got, want)
    assert helpers.bits_equal(la.cpu().numpy(), ext)                # ghosts of the input now hold the exchanged planes

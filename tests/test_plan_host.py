"""Host-side plan logic reached through the C ABI (pure host code: runs without a GPU).

Covers what the reference's verifier/lowering would accept or choke on:
  * result shape == input 0 shape            (DataflowLowering.cpp:285-286)
  * accesses leaving the input box are UB in the reference (smoke_apply.mlir:4-9) -> EOOB here
  * which kernel a geometry is routed to
"""
import ctypes as C

import pytest

from neptune_hip import _capi
from neptune_hip.geometry import interior_geom, make_geom

FAKE = 0x7F0000000000  # plan only inspects alignment, never dereferences


def _plan(lib, body, g, cfg=None, in_ptr=FAKE, out_ptr=FAKE + (1 << 36)):
    ins = (C.c_void_p * 1)(in_ptr)
    return lib.neptune_hip_apply_builtin_plan(body, C.byref(g), ins, out_ptr, C.byref(cfg) if cfg else None)


def _radius(*rows):
    r = ((C.c_int32 * 3) * 4)()
    for k in range(4):
        for d in range(3):
            r[k][d] = -1
    for k, row in enumerate(rows):
        for d, v in enumerate(row):
            r[k][d] = v
    return r


def test_fixture_geometries_plan_onto_the_march_kernel(built_libs):
    lib = _capi.load()
    assert _plan(lib, _capi.BODY_LAP3D7_F64, interior_geom((1024, 1024, 1024))) == _capi.KERNEL_MARCH
    assert _plan(lib, _capi.BODY_LAP3D7_F64, interior_geom((512, 512, 512))) == _capi.KERNEL_MARCH
    assert _plan(lib, _capi.BODY_LAP2D5_F64, interior_geom((8192, 8192))) == _capi.KERNEL_MARCH
    assert _plan(lib, _capi.BODY_LAP3D27_F32, interior_geom((512, 512, 512))) == _capi.KERNEL_MARCH
    # rows narrower than a wave go to the direct kernel
    assert _plan(lib, _capi.BODY_LAP1D3_F64, interior_geom((16,))) == _capi.KERNEL_DIRECT
    assert _plan(lib, _capi.BODY_LAP3D7_F64, interior_geom((16, 16, 16))) == _capi.KERNEL_DIRECT
    # odd last extent (node-centred 2^k+1 grids): ragged rows still march -- unaligned 16-byte accesses, the
    # last few cells of every row go to a direct-kernel launch -- unless what is left is narrower than a wave
    assert _plan(lib, _capi.BODY_LAP2D5_F64, interior_geom((100, 1023))) == _capi.KERNEL_MARCH
    assert _plan(lib, _capi.BODY_LAP3D7_F64, interior_geom((65, 65, 1025))) == _capi.KERNEL_MARCH
    assert _plan(lib, _capi.BODY_LAP3D7_F64, interior_geom((64, 64, 129))) == _capi.KERNEL_DIRECT   # 126 storable cells < 128
    assert _plan(lib, _capi.BODY_LAP3D27_F32, interior_geom((9, 9, 263))) == _capi.KERNEL_MARCH     # 256 storable
    assert _plan(lib, _capi.BODY_LAP3D27_F32, interior_geom((9, 9, 259))) == _capi.KERNEL_DIRECT    # 252 storable < 256


def test_forced_kernels(built_libs):
    lib = _capi.load()
    g = interior_geom((64, 64, 64))
    assert _plan(lib, _capi.BODY_LAP3D7_F64, g, _capi.LaunchCfg(_capi.KERNEL_MARCH, 0, 0, 0)) == _capi.KERNEL_MARCH
    assert _plan(lib, _capi.BODY_LAP3D7_F64, g, _capi.LaunchCfg(_capi.KERNEL_DIRECT, 0, 0, 0)) == _capi.KERNEL_DIRECT
    # march serves ragged rows (odd contiguous extent) as long as one whole lane vector is left to store ...
    odd = interior_geom((64, 64, 65))
    assert _plan(lib, _capi.BODY_LAP3D7_F64, odd, _capi.LaunchCfg(_capi.KERNEL_MARCH, 0, 0, 0)) == _capi.KERNEL_MARCH
    # ... but not rows too short for that, nor misaligned buffers: refused, never silently rerouted
    tiny = interior_geom((64, 64, 3))
    assert _plan(lib, _capi.BODY_LAP3D7_F64, tiny, _capi.LaunchCfg(_capi.KERNEL_MARCH, 0, 0, 0)) == _capi.EUNSUPPORTED
    assert _plan(lib, _capi.BODY_LAP3D7_F64, g, _capi.LaunchCfg(_capi.KERNEL_MARCH, 0, 0, 0),
                 in_ptr=FAKE + 8) == _capi.EUNSUPPORTED
    # auto with a misaligned buffer falls back to the direct kernel
    big = interior_geom((256, 256, 256))
    assert _plan(lib, _capi.BODY_LAP3D7_F64, big, None, in_ptr=FAKE + 8) == _capi.KERNEL_DIRECT


def test_region_restricting_inner_dims_uses_direct(built_libs):
    lib = _capi.load()
    n = (256, 256, 256)
    box = ([0, 0, 0], list(n))
    b = ([1, 1, 1], [255, 255, 255])
    planes = make_geom(box, b, region=([10, 0, 0], [20, 256, 256]))
    assert _plan(lib, _capi.BODY_LAP3D7_F64, planes) == _capi.KERNEL_MARCH
    sub = make_geom(box, b, region=([0, 8, 0], [256, 16, 256]))
    assert _plan(lib, _capi.BODY_LAP3D7_F64, sub) == _capi.KERNEL_DIRECT


def test_out_of_bounds_access_is_rejected_at_plan_time(built_libs):
    lib = _capi.load()
    # the reference's smoke_apply.mlir applies offsets +-1 over the FULL range [0,4): reads in[-1], in[4]
    full = make_geom(([0], [4]), ([0], [4]))
    assert lib.neptune_hip_check_geom(C.byref(full), C.byref(_radius((1, 0, 0)))) == _capi.EOOB
    assert _plan(lib, _capi.BODY_LAP1D3_F64, full) == _capi.EOOB
    ok = make_geom(([0], [16]), ([1], [15]))
    assert lib.neptune_hip_check_geom(C.byref(ok), C.byref(_radius((1, 0, 0)))) == _capi.OK
    # one cell too far on the upper side of dim 1
    g = make_geom(([0, 0, 0], [8, 8, 8]), ([1, 1, 1], [7, 8, 7]))
    assert lib.neptune_hip_check_geom(C.byref(g), C.byref(_radius((1, 1, 1)))) == _capi.EOOB
    # a pointwise body (radius 0) may cover the whole box
    g0 = make_geom(([0, 0], [8, 8]), ([0, 0], [8, 8]))
    assert lib.neptune_hip_check_geom(C.byref(g0), C.byref(_radius((0, 0, 0)))) == _capi.OK
    # empty bounds: nothing is accessed, nothing can be out of bounds
    ge = make_geom(([0], [4]), ([2], [2]))
    assert lib.neptune_hip_check_geom(C.byref(ge), C.byref(_radius((1, 0, 0)))) == _capi.OK


def test_logical_origins_shift_the_check(built_libs):
    lib = _capi.load()
    # input box [-1,17), result box [0,18) (same shape), bounds [0,16): accesses reach -1..16 -> inside
    g = make_geom(([0], [18]), ([0], [16]), in_boxes=[([-1], [17])])
    assert lib.neptune_hip_check_geom(C.byref(g), C.byref(_radius((1, 0, 0)))) == _capi.OK
    g2 = make_geom(([0], [18]), ([0], [17]), in_boxes=[([-1], [17])])
    assert lib.neptune_hip_check_geom(C.byref(g2), C.byref(_radius((1, 0, 0)))) == _capi.EOOB


def test_malformed_geometry(built_libs):
    lib = _capi.load()
    # result shape must equal input 0's shape
    g = make_geom(([0, 0], [8, 8]), ([1, 1], [7, 7]), in_boxes=[([0, 0], [8, 9])])
    assert lib.neptune_hip_check_geom(C.byref(g), None) == _capi.EINVAL
    # bounds outside the result box: the yield store would leave the buffer
    g = make_geom(([0, 0], [8, 8]), ([1, 1], [9, 7]))
    assert lib.neptune_hip_check_geom(C.byref(g), None) == _capi.EOOB
    g = interior_geom((8, 8))
    g.rank = 0
    assert lib.neptune_hip_check_geom(C.byref(g), None) == _capi.EINVAL
    g = interior_geom((8, 8))
    g.region_ub[0] = 9
    assert lib.neptune_hip_check_geom(C.byref(g), None) == _capi.EINVAL
    # wrong rank for the body
    assert _plan(lib, _capi.BODY_LAP3D7_F64, interior_geom((64, 64))) == _capi.EINVAL
    assert lib.neptune_hip_apply_builtin_plan(99, C.byref(interior_geom((8,))), (C.c_void_p * 1)(FAKE), FAKE, None) \
        == _capi.EINVAL


def test_store_argument_checks_are_host_side(built_libs):
    lib = _capi.load()
    a = lambda *v: (C.c_int64 * len(v))(*v)
    # box leaves the source buffer -> EOOB before anything is launched
    rc = lib.neptune_hip_store_box(_capi.F64, 1, FAKE, a(0), a(16), FAKE + 4096, a(0), a(16), a(8), a(17), None)
    assert rc == _capi.EOOB
    rc = lib.neptune_hip_store_box(_capi.F64, 4, FAKE, a(0), a(16), FAKE + 4096, a(0), a(16), a(0), a(16), None)
    assert rc == _capi.EINVAL
    assert lib.neptune_hip_store_full(7, FAKE, FAKE + 4096, 4, None) == _capi.EINVAL


def test_slab_plan_validation_needs_no_gpu(built_libs):
    """neptune_hip_slab_plan_create decides everything on the host: malformed requests come back NULL with a reason;
    a plan without ghost planes creates no stream and is destroyed again (no device involved)"""
    import ctypes as C
    from neptune_hip import _capi
    from neptune_hip.geometry import make_geom
    lib = _capi.load()
    box = ([0, 0, 0], [10, 8, 128])
    g = make_geom(box, ([1, 1, 1], [9, 7, 127]), [box])
    err = lambda: (lib.neptune_hip_slab_last_error() or b"").decode()
    make = lambda *a: lib.neptune_hip_slab_plan_create(*a)
    assert not make(None, None, 99, _capi.F64, C.byref(g), 1, 0, 0, -1, -1, None) and "built-in body" in err()
    assert not make(None, None, _capi.BODY_LAP3D7_F64, 7, C.byref(g), 1, 0, 0, -1, -1, None) and "element type" in err()
    # ghost planes need a communicator; a slab of 10 planes cannot hold 6 + 6 ghost planes
    assert not make(None, None, _capi.BODY_LAP3D7_F64, _capi.F64, C.byref(g), 1, 1, 1, 0, 0, None) and "communicator" in err()
    assert not make(None, None, _capi.BODY_LAP3D7_F64, _capi.F64, C.byref(g), 1, 6, 6, 0, 0, None)
    assert not make(None, None, _capi.BODY_LAP3D7_F64, _capi.F64, None, 1, 0, 0, -1, -1, None) and "geometry" in err()
    plan = make(None, None, _capi.BODY_LAP3D7_F64, _capi.F64, C.byref(g), 1, 0, 0, -1, -1, None)
    assert plan
    lib.neptune_hip_slab_plan_destroy(plan)
    assert lib.neptune_hip_halo_exchange(None, None, 0, 0, 0, 0, -1, -1, None) == _capi.EINVAL


def test_bounds_empty_along_one_dimension_are_zero_trips_wherever_they_lie(built_libs):
    """the reference's loop nest runs no iteration when ANY dimension of apply.bounds is empty
    (DataflowLowering.cpp:289-310), so bounds that are empty along one dimension and leave the result box -- or the
    footprint's reach -- along another are not an out-of-bounds plan"""
    import ctypes as C
    from neptune_hip import _capi
    from neptune_hip.geometry import make_geom
    lib = _capi.load()
    box = ([0, 0], [16, 128])
    radius = ((C.c_int32 * _capi.MAX_RANK) * _capi.MAX_INPUTS)()
    for k in range(_capi.MAX_INPUTS):
        for d in range(_capi.MAX_RANK):
            radius[k][d] = 1 if k == 0 and d < 2 else -1
    ok_geom = make_geom(box, ([1, 1], [15, 127]), [box])
    assert lib.neptune_hip_check_geom(C.byref(ok_geom), radius) == _capi.OK
    out_of_box = make_geom(box, ([1, 1], [15, 200]), [box])
    assert lib.neptune_hip_check_geom(C.byref(out_of_box), radius) == _capi.EOOB
    empty_and_out = make_geom(box, ([5, 1], [5, 200]), [box])          # dim 0 empty, dim 1 leaves the box
    assert lib.neptune_hip_check_geom(C.byref(empty_and_out), radius) == _capi.OK
    empty_and_reach = make_geom(box, ([0, 7], [16, 7]), [box])         # dim 1 empty, dim 0 would reach outside
    assert lib.neptune_hip_check_geom(C.byref(empty_and_reach), radius) == _capi.OK

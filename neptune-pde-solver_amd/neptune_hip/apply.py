"""Host-side launch of the apply hot path through the C ABI (neptune_hip_apply_builtin & co).

This is the thin layer the Python frontend, the tests and the bench share; it mirrors what the
emitted host C++ of a lowered module does for one `neptune_ir.apply` + `neptune_ir.store`.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

from . import _capi
from .fields import DeviceField, current_stream_ptr
from .geometry import Box, make_geom

BODY_BY_NAME = {
    "lap2d5_f64": _capi.BODY_LAP2D5_F64,
    "lap3d7_f64": _capi.BODY_LAP3D7_F64,
    "lap3d27_f32": _capi.BODY_LAP3D27_F32,
    "lap1d3_f64": _capi.BODY_LAP1D3_F64,
}
BODY_DTYPE = {_capi.BODY_LAP2D5_F64: _capi.F64, _capi.BODY_LAP3D7_F64: _capi.F64,
              _capi.BODY_LAP3D27_F32: _capi.F32, _capi.BODY_LAP1D3_F64: _capi.F64}
BODY_RANK = {_capi.BODY_LAP2D5_F64: 2, _capi.BODY_LAP3D7_F64: 3, _capi.BODY_LAP3D27_F32: 3,
             _capi.BODY_LAP1D3_F64: 1}
BODY_POINTS = {_capi.BODY_LAP2D5_F64: 5, _capi.BODY_LAP3D7_F64: 7, _capi.BODY_LAP3D27_F32: 27,
               _capi.BODY_LAP1D3_F64: 3}


def _in_array(inputs: Sequence[DeviceField]):
    arr = (C.c_void_p * len(inputs))(*[f.ptr for f in inputs])
    return arr


def make_cfg(kernel: int = _capi.KERNEL_AUTO, variant: int = -1, chunk: int = 0, flags: int = 0) -> _capi.LaunchCfg:
    """variant -1 = the library's default tile for the stencil shape; flags: _capi.FLAG_DIRECT_FLAT"""
    return _capi.LaunchCfg(kernel, variant, chunk, flags)


def geom_for(inputs: Sequence[DeviceField], out: DeviceField, bounds: Box, region: Optional[Box] = None):
    return make_geom(out.box, bounds, [f.box for f in inputs], region)


def apply_builtin(body, inputs: Sequence[DeviceField], out: DeviceField, bounds: Box,
                  region: Optional[Box] = None, cfg: Optional[_capi.LaunchCfg] = None,
                  stream: Optional[int] = None) -> None:
    """out = apply(inputs) {bounds}; asynchronous on `stream`.  `body`: a built-in body id, or the geometry-level
    entry of a lowered module's apply (neptune_hip.lowering.LoweredModule.geom_entry)."""
    lib = _capi.load()
    g = geom_for(inputs, out, bounds, region)
    st = current_stream_ptr() if stream is None else stream
    if hasattr(body, "fn"):
        rc = body(g, _in_array(inputs), out.ptr, st, cfg)
        _capi.check(rc, body.symbol)
        return
    rc = lib.neptune_hip_apply_builtin(body, C.byref(g), _in_array(inputs), out.ptr, st,
                                       C.byref(cfg) if cfg is not None else None)
    _capi.check(rc, "neptune_hip_apply_builtin")


def step_loop(body, a: DeviceField, b: DeviceField, bounds: Box, steps: int, others: Sequence[DeviceField] = (),
              cfg: Optional[_capi.LaunchCfg] = None, stream: Optional[int] = None) -> DeviceField:
    """`steps` applies in a row, ping-ponging between fields a and b (step 0 reads a); `others` are the fixed
    inputs 1.. of a multi-input body.  The pair of launches is captured once into a hipGraph and replayed, so small
    fields are not bound by launch overhead.  Asynchronous; returns the field holding the newest state."""
    lib = _capi.load()
    g = geom_for([a] + list(others), b, bounds)
    fields2 = (C.c_void_p * 2)(a.ptr, b.ptr)
    ins = _in_array([a] + list(others))
    st = current_stream_ptr() if stream is None else stream
    is_entry = hasattr(body, "fn")
    fn = C.cast(body.fn, C.c_void_p) if is_entry else None
    fn2 = C.cast(body.fn2, C.c_void_p) if is_entry and getattr(body, "fn2", None) is not None else None
    fn3 = C.cast(body.fn3, C.c_void_p) if is_entry and getattr(body, "fn3", None) is not None else None
    rc = lib.neptune_hip_step_loop_chain(fn, fn2, fn3, -1 if is_entry else body, C.byref(g), fields2, ins, steps, st,
                                         C.byref(cfg) if cfg is not None else None)
    _capi.check(rc, "neptune_hip_step_loop_chain")
    return b if steps % 2 else a


def apply_twice(body, inp: DeviceField, out: DeviceField, bounds: Box, region: Optional[Box] = None,
                cfg: Optional[_capi.LaunchCfg] = None, stream: Optional[int] = None, applies: int = 2) -> bool:
    """out = A(A(inp)) -- or A(A(A(inp))) with applies=3 -- in ONE pass over HBM for apply A (a built-in body id or a
    lowered apply's geometry-level entry); returns False -- having launched nothing -- when the body or the geometry does
    not qualify."""
    lib = _capi.load()
    g = geom_for([inp], out, bounds, region)
    st = current_stream_ptr() if stream is None else stream
    cfg_p = C.byref(cfg) if cfg is not None else None
    if hasattr(body, "fn"):
        f = getattr(body, "fn2" if applies == 2 else "fn3", None)
        if f is None:
            return False
        rc = f(C.byref(g), _in_array([inp]), out.ptr, st, cfg_p)
        what = body.symbol + str(applies)
    else:
        rc = lib.neptune_hip_apply_chain_builtin(body, applies, C.byref(g), _in_array([inp]), out.ptr, st, cfg_p)
        what = "neptune_hip_apply_chain_builtin"
    if rc == _capi.EUNSUPPORTED:
        return False
    _capi.check(rc, what)
    return True


def plan_builtin(body: int, inputs: Sequence[DeviceField], out: DeviceField, bounds: Box,
                 region: Optional[Box] = None, cfg: Optional[_capi.LaunchCfg] = None) -> int:
    lib = _capi.load()
    g = geom_for(inputs, out, bounds, region)
    return _capi.check(lib.neptune_hip_apply_builtin_plan(body, C.byref(g), _in_array(inputs), out.ptr,
                                                          C.byref(cfg) if cfg is not None else None),
                       "neptune_hip_apply_builtin_plan")


def time_builtin(body, inputs: Sequence[DeviceField], out: DeviceField, bounds: Box,
                 cfg: Optional[_capi.LaunchCfg] = None, warmup: int = 3, reps: int = 20,
                 stream: Optional[int] = None) -> float:
    """average milliseconds per launch, HIP events on the launch stream (blocking).  `body`: a built-in body id
    or a lowered apply's geometry-level entry (LoweredModule.geom_entry)."""
    lib = _capi.load()
    g = geom_for(inputs, out, bounds)
    st = current_stream_ptr() if stream is None else stream
    cfg_p = C.byref(cfg) if cfg is not None else None
    if hasattr(body, "fn"):
        what = "neptune_hip_time_apply_fn"
        ms = lib.neptune_hip_time_apply_fn(C.cast(body.fn, C.c_void_p), C.byref(g), _in_array(inputs), out.ptr, st, cfg_p,
                                           warmup, reps)
    else:
        what = "neptune_hip_time_apply_builtin"
        ms = lib.neptune_hip_time_apply_builtin(body, C.byref(g), _in_array(inputs), out.ptr, st, cfg_p, warmup, reps)
    if ms < 0:
        raise _capi.NeptuneHipError(int(ms), what)
    return ms


def store(src: DeviceField, dst: DeviceField, bounds: Optional[Box] = None, stream: Optional[int] = None) -> None:
    """neptune_ir.store %src to %dst {bounds?}  (DataflowLowering.cpp:165-220)"""
    lib = _capi.load()
    st = current_stream_ptr() if stream is None else stream
    if src.dtype != dst.dtype:
        raise ValueError("store: element types differ")
    if bounds is None:
        if src.shape != dst.shape:
            raise ValueError("store: shapes differ")
        _capi.check(lib.neptune_hip_store_full(src.dtype, src.ptr, dst.ptr, src.count, st), "neptune_hip_store_full")
        return
    r = src.rank
    arr = lambda v: (C.c_int64 * r)(*[int(x) for x in v])
    _capi.check(lib.neptune_hip_store_box(src.dtype, r, src.ptr, arr(src.lb), arr(src.ub), dst.ptr, arr(dst.lb),
                                          arr(dst.ub), arr(bounds[0]), arr(bounds[1]), st), "neptune_hip_store_box")


def count_mismatch(a: DeviceField, b: DeviceField) -> int:
    lib = _capi.load()
    if a.dtype != b.dtype or a.shape != b.shape:
        raise ValueError("count_mismatch: fields differ in type or shape")
    n = lib.neptune_hip_count_mismatch(a.dtype, a.ptr, b.ptr, a.count, current_stream_ptr())
    if n < 0:
        raise RuntimeError("neptune_hip_count_mismatch failed")
    return int(n)


def reduce_sum(src: DeviceField, bounds: Optional[Box] = None, stream: Optional[int] = None) -> float:
    """neptune_ir.reduce %src (in bounds)? {kind = "sum"}  (DataflowLowering.cpp:589-698); blocking"""
    lib = _capi.load()
    r = src.rank
    arr = lambda v: (C.c_int64 * r)(*[int(x) for x in v])
    out = C.c_double(0.0)
    rc = lib.neptune_hip_reduce_sum(src.dtype, r, src.ptr, arr(src.lb), arr(src.ub),
                                    arr(bounds[0]) if bounds is not None else None,
                                    arr(bounds[1]) if bounds is not None else None, C.byref(out),
                                    current_stream_ptr() if stream is None else stream)
    _capi.check(rc, "neptune_hip_reduce_sum")
    return out.value


def autotune_builtin(body, inputs: Sequence[DeviceField], out: DeviceField, bounds: Box,
                     region: Optional[Box] = None, reps: int = 5, stream: Optional[int] = None):
    """plan-time tuning: -> (LaunchCfg of the fastest tile/chunk for this geometry, its ms per launch).  `body`: a
    built-in body id (every tile of the library is a candidate) or a lowered apply's geometry-level entry (the tiles
    its module holds: GeomEntry.num_variants)."""
    lib = _capi.load()
    g = geom_for(inputs, out, bounds, region)
    best = _capi.LaunchCfg(0, -1, 0, 0)
    ms = C.c_double(0.0)
    st = current_stream_ptr() if stream is None else stream
    if hasattr(body, "fn"):
        rc = lib.neptune_hip_autotune_fn(C.cast(body.fn, C.c_void_p), body.num_variants, C.byref(g), _in_array(inputs),
                                         out.ptr, st, reps, C.byref(best), C.byref(ms))
        _capi.check(rc, "neptune_hip_autotune_fn")
    else:
        rc = lib.neptune_hip_autotune_builtin(body, C.byref(g), _in_array(inputs), out.ptr, st, reps, C.byref(best),
                                              C.byref(ms))
        _capi.check(rc, "neptune_hip_autotune_builtin")
    return best, ms.value

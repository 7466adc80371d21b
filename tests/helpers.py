"""Shared test helpers: oracle access, deterministic fields, fixture texts.

Only tests (and __graft_entry__.smoke / bench.py's cpu_baseline leg) may touch oracle/."""
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
REFERENCE_SMOKE = REPO / "tests" / "golden"  # the reference tree itself is absent on the GPU box
sys.path.insert(0, str(REPO / "oracle"))
sys.path.insert(0, str(REPO / "tools"))

import neptune_oracle as oracle  # noqa: E402
from make_stencil_mlir import KINDS, stencil_module  # noqa: E402

FIXTURE_DIR = REPO / "tests" / "mlir_tests" / "conversion_tests"
GOLDEN_DIR = REPO / "tests" / "golden"

_MASK = (1 << 64) - 1


def _mix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def hash_field(shape, dtype=np.float64, seed=1, index_offset=0):
    """numpy twin of neptune_hip_fill_hash (csrc/kernels/apply_common.hpp: hash_f64 / hash_f32)"""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        idx = (np.arange(n, dtype=np.int64) + np.int64(index_offset)).astype(np.uint64)
        ms = _mix64(np.uint64(seed))
        h = _mix64(idx ^ ms)
    if np.dtype(dtype) == np.float64:
        v = (h >> np.uint64(12)).astype(np.float64) * (1.0 / 2251799813685248.0) - 1.0
    else:
        v = (h >> np.uint64(41)).astype(np.float32) * np.float32(1.0 / 4194304.0) - np.float32(1.0)
    return v.reshape(shape)


def bits_equal(a: np.ndarray, b: np.ndarray) -> bool:
    """bit-for-bit equality (NaN-safe, distinguishes -0.0)"""
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    u = {8: np.uint64, 4: np.uint32}[a.dtype.itemsize]
    return bool(np.array_equal(np.ascontiguousarray(a).view(u), np.ascontiguousarray(b).view(u)))


def mismatch_report(a: np.ndarray, b: np.ndarray, limit=5) -> str:
    u = {8: np.uint64, 4: np.uint32}[a.dtype.itemsize]
    bad = np.argwhere(np.ascontiguousarray(a).view(u) != np.ascontiguousarray(b).view(u))
    lines = [f"{len(bad)} mismatching cells of {a.size}"]
    for idx in bad[:limit]:
        t = tuple(int(x) for x in idx)
        lines.append(f"  {t}: got {a[t]!r} expected {b[t]!r}")
    return "\n".join(lines)


def oracle_module(kind, shape, origin=None, bounds=None):
    return oracle.Module.parse(stencil_module(kind, shape, origin, bounds))


def oracle_entry(kind, u: np.ndarray, origin=None, bounds=None) -> np.ndarray:
    """result of the fixture's @entry(out, in) on input u, per the numpy oracle"""
    m = oracle_module(kind, u.shape, origin, bounds)
    out = np.zeros_like(u)
    r = m.call("entry", out, u)
    assert r is out
    return out


def load_liboracle():
    path = REPO / "oracle" / "_build" / "liboracle.so"
    lib = C.CDLL(str(path))
    i64p = C.POINTER(C.c_int64)
    dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
    for name, ptr, nd in (("lap2d5_f64", dp, 2), ("lap3d7_f64", dp, 3), ("lap3d27_f32", fp, 3), ("lap1d3_f64", dp, 1)):
        for pre in ("ref_entry_", "ref_fused_"):
            if pre == "ref_fused_" and nd == 1:
                continue
            fn = getattr(lib, pre + name)
            fn.restype = C.c_int
            fn.argtypes = [ptr, ptr] + [C.c_int64] * nd + [i64p, i64p]
    lib.ref_fill_hash_f64.argtypes = [dp, C.c_int64, C.c_int64, C.c_uint64]
    lib.ref_fill_hash_f32.argtypes = [fp, C.c_int64, C.c_int64, C.c_uint64]
    lib.ref_fill_hash_f64.restype = None
    lib.ref_fill_hash_f32.restype = None
    lib.ref_num_threads.restype = C.c_int
    return lib


C_NAME = {"2d5": "lap2d5_f64", "3d7": "lap3d7_f64", "3d27": "lap3d27_f32", "1d3": "lap1d3_f64"}


def c_oracle_entry(kind, u: np.ndarray, variant="entry", lb=None, ub=None) -> np.ndarray:
    lib = load_liboracle()
    fn = getattr(lib, f"ref_{variant}_{C_NAME[kind]}")
    u = np.ascontiguousarray(u)
    out = np.zeros_like(u)
    nd = u.ndim
    lb = [1] * nd if lb is None else list(lb)
    ub = [n - 1 for n in u.shape] if ub is None else list(ub)
    ct = C.c_double if u.dtype == np.float64 else C.c_float
    rc = fn(out.ctypes.data_as(C.POINTER(ct)), u.ctypes.data_as(C.POINTER(ct)), *[C.c_int64(n) for n in u.shape],
            (C.c_int64 * nd)(*lb), (C.c_int64 * nd)(*ub))
    assert rc == 0
    return out


def load_kats():
    return json.loads((GOLDEN_DIR / "kat_reference_smoke.json").read_text())


def prefetch_modules(texts, workers=None):
    """compile (lower + hipcc) a batch of module texts into the current module cache with a pool of host threads, so
    that the tests that follow load them as cache hits: hipcc takes 3-10 s per module on one core and the GPU box has
    16 of them.  The lowering is called through ctypes (the GIL is released for the whole compile); nothing is loaded,
    nothing touches the GPU.  A text that fails to compile is left to its test to report."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    from neptune_hip import lowering
    texts = list(dict.fromkeys(texts))
    if not texts:
        return
    if workers is None:
        workers = max(1, min(12, (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count() or 1) - 2))

    def one(text):
        try:
            lowering.compile_module(text, load=False)
        except Exception:       # noqa: BLE001 - the test that needs this module shows the diagnostic
            pass
    with ThreadPoolExecutor(max_workers=workers) as pool:
        list(pool.map(one, texts))

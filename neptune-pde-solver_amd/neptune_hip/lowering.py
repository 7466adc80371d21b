"""Python face of the NeptuneIR -> HIP lowering (libneptune_lowering.so) and loader of the
modules it produces.

Mirrors the reference's AOT path (python_frontend/neptune/backend.py:15-75): the module text is
hashed (sha256, first 16 hex digits), the shared object is cached as
$NEPTUNE_CACHE_DIR/neptune_kernel_<hash>.so (default ~/.neptune/cache) and loaded with ctypes.
A lowered module exports the reference's calling convention: expanded memref arguments, memref
struct result (test/smoke_tests/smoke_apply.sh:39-50)."""
from __future__ import annotations

import ctypes as C
import hashlib
import json
import os
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _capi

LOWERING_LIB = _capi.PKG_ROOT / "lib" / "libneptune_lowering.so"
_NP = {"f64": np.float64, "f32": np.float32}
_lib = None


class LoweringError(RuntimeError):
    pass


def _load():
    global _lib
    if _lib is None:
        if not LOWERING_LIB.exists():
            raise ImportError(f"{LOWERING_LIB} not found: build it with `make lowering`")
        lib = C.CDLL(str(LOWERING_LIB))
        cpp = C.POINTER(C.c_char_p)
        lib.neptune_lowering_verify.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        lib.neptune_lowering_to_hip.argtypes = [C.c_char_p] + [C.POINTER(C.c_void_p)] * 3
        lib.neptune_lowering_compile.argtypes = [C.c_char_p] * 4 + [C.POINTER(C.c_void_p)] * 2
        lib.neptune_lowering_free.argtypes = [C.c_void_p]
        lib.neptune_lowering_version.restype = C.c_char_p
        for f in (lib.neptune_lowering_verify, lib.neptune_lowering_to_hip, lib.neptune_lowering_compile):
            f.restype = C.c_int
        del cpp
        _lib = lib
    return _lib


def _take(lib, p: C.c_void_p) -> Optional[str]:
    if not p.value:
        return None
    s = C.string_at(p.value).decode()
    lib.neptune_lowering_free(p)
    return s


def verify(text: str) -> None:
    """raise LoweringError with the reference's diagnostic if the module is ill-formed"""
    lib = _load()
    diag = C.c_void_p()
    if lib.neptune_lowering_verify(text.encode(), C.byref(diag)) != 0:
        raise LoweringError(_take(lib, diag))


def to_hip(text: str):
    """-> (HIP source, report dict)"""
    lib = _load()
    src, rep, diag = C.c_void_p(), C.c_void_p(), C.c_void_p()
    if lib.neptune_lowering_to_hip(text.encode(), C.byref(src), C.byref(rep), C.byref(diag)) != 0:
        raise LoweringError(_take(lib, diag))
    return _take(lib, src), json.loads(_take(lib, rep))


def cache_dir() -> Path:
    env = os.environ.get("NEPTUNE_CACHE_DIR")
    d = Path(env) if env else Path.home() / ".neptune" / "cache"
    d.mkdir(parents=True, exist_ok=True)
    return d


_build_id = None


def build_id() -> str:
    """identifies everything besides the module text that a compiled module bakes in: the emitter
    (libneptune_lowering.so), the header-only kernels and runtime it is compiled against, the public ABI header,
    and the runtime library it links (its contents: the ABI -- geometry struct, pool, tile numbering -- moves with
    the sources, not with the file's time stamp, so a `make` that rebuilds the same library or a copy of the tree onto
    another machine keeps the cache and the stored launch choices valid).  A cached shared object built against any
    other state is never loaded."""
    global _build_id
    if _build_id is None:
        h = hashlib.sha256()
        h.update(_load().neptune_lowering_version())
        csrc = _capi.PKG_ROOT / "csrc"
        files = sorted((csrc / "kernels").glob("*.hpp")) + sorted((csrc / "runtime").glob("*.hpp"))
        files += [_capi.HEADER_PATH, LOWERING_LIB]
        for f in files:
            h.update(f.name.encode())
            h.update(f.read_bytes())
        rt = _capi.library_path()
        if rt.exists():
            h.update(rt.name.encode())
            with open(rt, "rb") as f:
                while True:
                    block = f.read(1 << 22)
                    if not block:
                        break
                    h.update(block)
        _build_id = h.hexdigest()[:16]
    return _build_id


def module_hash(text: str) -> str:
    """cache key: sha256 over the module text (reference contract, backend.py:26-41: sha256(IR)[:16]) AND this
    build's id -- the reference's artefact depends on the IR alone, a lowered module also bakes in the kernels it was
    compiled against.  A build with every march tile compiled in (NEPTUNE_HIP_FULL_VARIANTS=1) is a different
    artefact of the same text."""
    full = os.environ.get("NEPTUNE_HIP_FULL_VARIANTS", "") not in ("", "0")
    key = text + ("\n// all march tiles" if full else "") + "\n// build " + build_id()
    return hashlib.sha256(key.encode("utf-8")).hexdigest()[:16]


def compile_module(text: str, so_path: Optional[os.PathLike] = None, use_cache: bool = True,
                   cache_directory: Optional[os.PathLike] = None, load: bool = True) -> Optional["LoweredModule"]:
    """lower + hipcc (gfx950) + load.  Compiling needs no GPU; loading needs libneptune_hip.so.  Without an explicit
    so_path the object lives in `cache_directory` (default: cache_dir()) under its module_hash.  load=False only fills
    the cache (what the profiling scripts do before they start rocprofv3: hipcc is started with an environment scrubbed
    of LD_PRELOAD / ROCP* / HSA_TOOLS_*, csrc/lowering/capi.cpp, but a profiled run should be a pure cache hit)."""
    lib = _load()
    if so_path is None:
        directory = Path(cache_directory) if cache_directory else cache_dir()
        directory.mkdir(parents=True, exist_ok=True)
        so_path = directory / f"neptune_kernel_{module_hash(text)}.so"
    so_path = Path(so_path)
    rep_path = so_path.with_suffix(".json")
    if not (use_cache and so_path.exists() and rep_path.exists()):
        rep, diag = C.c_void_p(), C.c_void_p()
        rc = lib.neptune_lowering_compile(text.encode(), str(so_path).encode(), str(_capi.REPO_ROOT).encode(),
                                          os.environ.get("HIPCC", "").encode() or None, C.byref(rep), C.byref(diag))
        if rc != 0:
            raise LoweringError(_take(lib, diag))
        rep_path.write_text(_take(lib, rep))
    if not load:
        return None
    return LoweredModule(so_path, json.loads(rep_path.read_text()))


class GeomEntry:
    """one apply of a lowered module, callable with an explicit geometry (see LoweredModule.geom_entry)"""

    def __init__(self, module: "LoweredModule", info: dict):
        self.module = module            # keeps the shared object loaded
        self.info = info
        self.symbol = info["geom_symbol"]
        self.rank, self.num_inputs, self.halo0 = info["rank"], info["inputs"], info["halo0"]
        self.dtype = _capi.F64 if info["elem"] == "f64" else _capi.F32
        self.fn = getattr(module.lib, self.symbol)
        self.fn.restype = C.c_int
        self.fn.argtypes = [C.POINTER(_capi.ApplyGeom), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p,
                            C.POINTER(_capi.LaunchCfg)]
        # march tiles the module holds for this entry (the library's defaults unless built with
        # NEPTUNE_HIP_FULL_VARIANTS=1): the candidates of neptune_hip.apply.autotune_builtin
        count = getattr(module.lib, self.symbol + "_variants")
        count.restype, count.argtypes = C.c_int, [C.c_int]
        self.num_variants = int(count(self.rank))
        # two chained applies in one pass over HBM (csrc/kernels/apply_march2.hpp); returns NEPTUNE_HIP_EUNSUPPORTED for
        # footprints / geometries it cannot take, and step loops then fall back to one launch per step
        self.fn2 = getattr(module.lib, self.symbol + "2", None)
        self.fn3 = getattr(module.lib, self.symbol + "3", None)
        for f in (self.fn2, self.fn3):
            if f is not None:
                f.restype, f.argtypes = C.c_int, self.fn.argtypes

    def __call__(self, geom, in_array, out_ptr, stream, cfg=None) -> int:
        return self.fn(C.byref(geom), in_array, out_ptr, stream, C.byref(cfg) if cfg is not None else None)


class LoweredModule:
    """a compiled module: call exported symbols with numpy arrays (host buffers: staged through the
    device, result comes back in malloc'ed host memory) or with DeviceField / torch CUDA tensors
    (device buffers: used in place, result stays on the device)"""

    def __init__(self, so_path: Path, report: dict):
        _capi.load()  # libneptune_hip.so first (RTLD_GLOBAL): the module links against it
        self.path = Path(so_path)
        self.report = report
        self.lib = C.CDLL(str(so_path))
        self.signatures: Dict[str, dict] = {s["name"]: s for s in report["signatures"]}
        for name, sig in self.signatures.items():
            fn = getattr(self.lib, name)
            argtypes: List = []
            for a in sig["args"]:
                argtypes += [C.c_void_p, C.c_void_p, C.c_int64] + [C.c_int64] * (2 * a["rank"])
            fn.argtypes = argtypes
            res = sig["result"]
            if res and res["kind"] == "scalar":
                fn.restype = C.c_double if res["elem"] == "f64" else C.c_float
            else:
                fn.restype = _capi.MEMREF[res["rank"]] if res else None

    @property
    def symbols(self) -> List[str]:
        return list(self.signatures)

    def geom_entry(self, function: str, index: int = 0) -> "GeomEntry":
        """geometry-level entry of the `index`-th apply of @function: the module's counterpart of
        neptune_hip_apply_builtin (caller-supplied boxes / bounds / region / stream / launch cfg).  Accepted
        wherever the built-in body ids are: neptune_hip.apply.apply_builtin, plan-free region launches,
        neptune_hip.slab.ShardedApply."""
        cands = [a for a in self.report["applies"] if a["function"] == function and a.get("geom_symbol")]
        if index >= len(cands):
            raise KeyError(f"@{function} has no apply #{index} with a geometry-level entry")
        return GeomEntry(self, cands[index])

    def call(self, name: str, *args):
        sig = self.signatures[name]
        if len(args) != len(sig["args"]):
            raise TypeError(f"@{name} takes {len(sig['args'])} arguments")
        flat: List = []
        keep = []
        device_mode = None
        for a, spec in zip(args, sig["args"]):
            if isinstance(a, np.ndarray):
                if a.dtype != _NP[spec["elem"]] or a.ndim != spec["rank"] or not a.flags["C_CONTIGUOUS"]:
                    raise TypeError(f"@{name}: argument must be a C-contiguous {spec['elem']} array of rank {spec['rank']}")
                ptr, shape, is_dev = a.ctypes.data, a.shape, False
            else:
                t = getattr(a, "tensor", a)  # DeviceField or torch tensor
                if not t.is_cuda or not t.is_contiguous() or t.dim() != spec["rank"]:
                    raise TypeError(f"@{name}: device argument must be a contiguous CUDA tensor of rank {spec['rank']}")
                ptr, shape, is_dev = t.data_ptr(), tuple(t.shape), True
            if device_mode is None:
                device_mode = is_dev
            keep.append(a)
            strides = [1] * len(shape)
            for d in range(len(shape) - 2, -1, -1):
                strides[d] = strides[d + 1] * shape[d + 1]
            flat += [ptr, ptr, 0] + list(shape) + strides
        fn = getattr(self.lib, name)
        ret = fn(*flat)
        if not sig["result"]:
            return None
        if sig["result"]["kind"] == "scalar":
            return float(ret)
        shape = tuple(ret.sizes)
        for a in keep:  # result aliases an argument (e.g. @entry returns its destination field)
            p = a.ctypes.data if isinstance(a, np.ndarray) else getattr(a, "tensor", a).data_ptr()
            if ret.aligned == p:
                return a
        hip = _capi.load()
        dt = _NP[sig["result"]["elem"]]
        count = int(np.prod(shape))
        if hip.neptune_hip_is_device_ptr(ret.aligned):
            import torch
            out = torch.empty(shape, dtype={np.float64: torch.float64, np.float32: torch.float32}[dt], device="cuda")
            hip.neptune_hip_memcpy_d2d(out.data_ptr(), ret.aligned, count * dt().itemsize, None)
            hip.neptune_hip_device_sync()
            hip.neptune_rt_free(ret.allocated)   # callee allocated, caller frees
            return out
        buf = (C.c_char * (count * dt().itemsize)).from_address(ret.aligned)
        out = np.frombuffer(buf, dtype=dt).reshape(shape).copy()
        hip.neptune_rt_free(ret.allocated)       # host result: free()
        return out

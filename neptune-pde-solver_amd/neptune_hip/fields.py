"""Device-resident fields: the run-time counterpart of `!neptune_ir.field` / `!neptune_ir.temp`.

A field is a dense row-major buffer of shape ub-lb for a logical box [lb,ub)
(reference: lib/Passes/DataflowLowering.cpp:41-49).  Device memory comes from torch (plumbing
only: allocation, H2D/D2H copies, streams); all arithmetic happens in libneptune_hip.so.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _capi

_TORCH_DTYPE = {_capi.F64: torch.float64, _capi.F32: torch.float32}
_NP_DTYPE = {_capi.F64: np.float64, _capi.F32: np.float32}
_FROM_NP = {np.dtype(np.float64): _capi.F64, np.dtype(np.float32): _capi.F32}
ELEM_NAME = {_capi.F64: "f64", _capi.F32: "f32"}
ELEM_SIZE = {_capi.F64: 8, _capi.F32: 4}


def require_gpu() -> None:
    """The product path has no CPU fallback: fail loudly without a HIP device."""
    if not torch.cuda.is_available():
        raise RuntimeError("NeptuneIR HIP backend: no HIP device visible (torch.cuda.is_available() is False)")
    lib = _capi.load()
    if not lib.neptune_hip_available():
        raise RuntimeError("NeptuneIR HIP backend: libneptune_hip.so sees no HIP device")


def current_stream_ptr() -> int:
    return int(torch.cuda.current_stream().cuda_stream)


class DeviceField:
    """dense row-major device buffer + its logical box"""

    def __init__(self, lb: Sequence[int], ub: Sequence[int], dtype: int = _capi.F64,
                 tensor: Optional[torch.Tensor] = None, device: Optional[torch.device] = None):
        self.lb = tuple(int(x) for x in lb)
        self.ub = tuple(int(x) for x in ub)
        if len(self.lb) != len(self.ub) or not (1 <= len(self.lb) <= _capi.MAX_RANK):
            raise ValueError("field rank must be 1..3")
        self.dtype = dtype
        shape = tuple(u - l for l, u in zip(self.lb, self.ub))
        if any(n <= 0 for n in shape):
            raise ValueError("empty field box")
        if tensor is None:
            require_gpu()
            tensor = torch.empty(shape, dtype=_TORCH_DTYPE[dtype], device=device or torch.device("cuda"))
        if tuple(tensor.shape) != shape or tensor.dtype != _TORCH_DTYPE[dtype] or not tensor.is_contiguous():
            raise ValueError("tensor does not match the field's box / dtype / dense row-major layout")
        self.tensor = tensor

    # ---- construction --------------------------------------------------------------------
    @classmethod
    def from_numpy(cls, a: np.ndarray, lb: Optional[Sequence[int]] = None) -> "DeviceField":
        a = np.ascontiguousarray(a)
        if a.dtype not in _FROM_NP:
            raise ValueError(f"unsupported element type {a.dtype}")
        lb = tuple(lb) if lb is not None else (0,) * a.ndim
        require_gpu()
        t = torch.from_numpy(a).to("cuda")
        return cls(lb, tuple(l + n for l, n in zip(lb, a.shape)), _FROM_NP[a.dtype], t)

    @classmethod
    def empty_like(cls, other: "DeviceField") -> "DeviceField":
        return cls(other.lb, other.ub, other.dtype, device=other.tensor.device)

    @classmethod
    def hashed(cls, shape: Sequence[int], dtype: int = _capi.F64, seed: int = 1, index_offset: int = 0,
               lb: Optional[Sequence[int]] = None) -> "DeviceField":
        """deterministic pseudo-random field in [-1,1), generated on the device
        (host twin: neptune_hip_hash_value / tests.helpers.hash_field)"""
        lb = tuple(lb) if lb is not None else (0,) * len(shape)
        f = cls(lb, tuple(l + n for l, n in zip(lb, shape)), dtype)
        f.fill_hash(seed, index_offset)
        return f

    def fill_hash(self, seed: int, index_offset: int = 0) -> None:
        lib = _capi.load()
        _capi.check(lib.neptune_hip_fill_hash(self.dtype, self.ptr, self.count, index_offset, seed,
                                              current_stream_ptr()), "neptune_hip_fill_hash")

    # ---- views ---------------------------------------------------------------------------
    @property
    def shape(self) -> Tuple[int, ...]:
        return tuple(self.tensor.shape)

    @property
    def rank(self) -> int:
        return len(self.lb)

    @property
    def count(self) -> int:
        return int(self.tensor.numel())

    @property
    def nbytes(self) -> int:
        return self.count * ELEM_SIZE[self.dtype]

    @property
    def ptr(self) -> int:
        return int(self.tensor.data_ptr())

    @property
    def box(self):
        return (self.lb, self.ub)

    def numpy(self) -> np.ndarray:
        return self.tensor.detach().cpu().numpy()

    def planes(self, lo: int, hi: int) -> np.ndarray:
        """download physical planes [lo,hi) along dim 0 only (full-size parity checks)"""
        return self.tensor[lo:hi].detach().cpu().numpy()

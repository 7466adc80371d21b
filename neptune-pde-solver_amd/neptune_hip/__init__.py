"""neptune_hip -- Python host layer over libneptune_hip.so, the MI355X backend for NeptuneIR's
stencil hot path (neptune_ir.apply / access / load / store).  See include/neptune_hip.h for the
C ABI and DESIGN.md for the layout and kernels."""
from . import _capi  # noqa: F401
from ._capi import (BODY_LAP1D3_F64, BODY_LAP2D5_F64, BODY_LAP3D7_F64, BODY_LAP3D27_F32, F32, F64,  # noqa: F401
                    KERNEL_AUTO, KERNEL_DIRECT, KERNEL_MARCH, NeptuneHipError)
from .geometry import interior_geom, make_geom  # noqa: F401

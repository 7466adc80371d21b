import sys, time
sys.path.insert(0, "/root/repo/neptune-pde-solver_amd")
import torch
from neptune_hip import _capi, apply, fields
lib = _capi.load(); lib.neptune_hip_init(0)
n = 1 << 27
a = fields.DeviceField.hashed((n,), _capi.F64, seed=1); b = fields.DeviceField.empty_like(a)
for name, k in (("direct", _capi.KERNEL_DIRECT), ("march", _capi.KERNEL_MARCH)):
    ms = apply.time_builtin(_capi.BODY_LAP1D3_F64, [a], b, ([1], [n - 1]), apply.make_cfg(k), 3, 20)
    print(f"1-D 3-pt n=2^27 f64 {name}: {ms:.4f} ms  {2*a.nbytes/ms/1e6:.0f} GB/s")

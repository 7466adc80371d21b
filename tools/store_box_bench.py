import sys
sys.path.insert(0, "neptune-pde-solver_amd")
import torch
from neptune_hip import _capi, apply, fields
lib = _capi.load(); lib.neptune_hip_init(0)
n = 1024
a = fields.DeviceField.hashed((n, n, n), _capi.F64, seed=1)
b = fields.DeviceField.empty_like(a)
box = ([1, 1, 1], [n - 1, n - 1, n - 1])
for _ in range(5): apply.store(a, b, box)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): apply.store(a, b, box)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"store_box interior of 1024^3 f64: {ms:.4f} ms {2 * (n - 2) ** 3 * 8 / ms / 1e6:.1f} GB/s")

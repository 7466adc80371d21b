import json, sys
sys.path.insert(0, "neptune-pde-solver_amd")
import torch
from neptune_hip import _capi, apply, fields
lib = _capi.load(); lib.neptune_hip_init(0)
body = apply.BODY_BY_NAME["lap3d7_f64"]
for shape in [(1025,)*3, (513,)*3, (769,)*3, (641,)*3, (385,)*3, (1024,)*3, (512,)*3]:
    a = fields.DeviceField.hashed(shape, _capi.F64, seed=3); b = fields.DeviceField.empty_like(a)
    bounds = ([1]*3, [n-1 for n in shape]); nbytes = 2*a.tensor.numel()*8
    row = {"shape": shape}
    for name, cfg in [("auto", None)] + [(f"c{c}", apply.make_cfg(_capi.KERNEL_MARCH, -1, c)) for c in (128, 96, 64, 48, 32)]:
        apply.time_builtin(body, [a], b, bounds, cfg=cfg, warmup=8, reps=8)
        ms = min(apply.time_builtin(body, [a], b, bounds, cfg=cfg, warmup=2, reps=20) for _ in range(2))
        row[name] = round(nbytes/ms/1e6)
    print(json.dumps(row), flush=True)
    del a, b; torch.cuda.empty_cache()

import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "neptune-pde-solver_amd"))
import torch
from neptune_hip import _capi, apply, fields
lib = _capi.load(); lib.neptune_hip_init(0)
for shape, dt in (((1024,1024,1024), _capi.F64), ((512,512,512), _capi.F32), ((8192,8192), _capi.F64)):
    f = fields.DeviceField.hashed(shape, dt, seed=3)
    apply.reduce_sum(f)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): r = apply.reduce_sum(f)
    dtm = (time.perf_counter() - t0) / n
    print(shape, "f64" if dt == _capi.F64 else "f32", f"{dtm*1e3:.3f} ms  {f.nbytes/dtm/1e9:.0f} GB/s  sum={r}")
    sub = ([1]*len(shape), [s-1 for s in shape])
    apply.reduce_sum(f, sub); t0 = time.perf_counter()
    for _ in range(n): r = apply.reduce_sum(f, sub)
    dtm = (time.perf_counter() - t0) / n
    print("   interior box", f"{dtm*1e3:.3f} ms  {f.nbytes/dtm/1e9:.0f} GB/s")
    del f; torch.cuda.empty_cache()

import sys, time
sys.path.insert(0, "/root/repo/neptune-pde-solver_amd")
import torch
from neptune_hip import _capi, apply, fields, slab as slab_mod
lib = _capi.load(); lib.neptune_hip_init(0)
for wl, body, shape, dt in (("3d27_512", _capi.BODY_LAP3D27_F32, (512,512,512), _capi.F32), ("3d7_512", _capi.BODY_LAP3D7_F64, (512,512,512), _capi.F64)):
    a = fields.DeviceField.hashed(shape, dt, seed=1); b = fields.DeviceField.empty_like(a)
    bounds = ([1]*3, [n-1 for n in shape])
    cfg = apply.make_cfg()
    for rep in range(2):
        ms_c = apply.time_builtin(body, [a], b, bounds, cfg, 3, 50)
        # python loop
        st = fields.current_stream_ptr()
        e0, e1 = lib.neptune_hip_event_create(), lib.neptune_hip_event_create()
        for _ in range(3): apply.apply_builtin(body, [a], b, bounds, cfg=cfg)
        torch.cuda.synchronize()
        lib.neptune_hip_event_record(e0, st)
        t0 = time.perf_counter()
        for _ in range(50): apply.apply_builtin(body, [a], b, bounds, cfg=cfg)
        t_issue = time.perf_counter() - t0
        lib.neptune_hip_event_record(e1, st); torch.cuda.synchronize()
        ms_py = lib.neptune_hip_event_elapsed_ms(e0, e1) / 50
        sl = slab_mod.decompose(([0]*3, list(shape)), 1, 0, 1)
        op = slab_mod.ShardedApply(sl, body, bounds, cfg=cfg)
        for _ in range(3): op(a, b)
        torch.cuda.synchronize()
        lib.neptune_hip_event_record(e0, st)
        for _ in range(50): op(a, b)
        lib.neptune_hip_event_record(e1, st); torch.cuda.synchronize()
        ms_op = lib.neptune_hip_event_elapsed_ms(e0, e1) / 50
        print(wl, f"C-loop {ms_c:.4f} ms | python apply_builtin {ms_py:.4f} ms (issue {t_issue/50*1e3:.4f} ms/launch) | ShardedApply {ms_op:.4f} ms")

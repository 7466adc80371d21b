#!/usr/bin/env python3
"""Ragged grids (2^k+1) on the wider tiles of the library: 1025^3 and 513^3 fp64 7-point, a few tiles x chunk lengths.
usage: tools/ragged_tiles.py"""
import sys
sys.path.insert(0, "neptune-pde-solver_amd")
import torch
from neptune_hip import _capi, apply, fields
lib = _capi.load(); lib.neptune_hip_init(0)
for body_name, shape in (("lap3d7_f64", (1025, 1025, 1025)), ("lap3d7_f64", (513, 513, 513))):
    body = apply.BODY_BY_NAME[body_name]
    a = fields.DeviceField.hashed(shape, apply.BODY_DTYPE[body], seed=3)
    b = fields.DeviceField.empty_like(a)
    bounds = ([1] * 3, [n - 1 for n in shape])
    nbytes = 2 * a.tensor.numel() * a.tensor.element_size()
    apply.time_builtin(body, [a], b, bounds, cfg=None, warmup=10, reps=5)
    for v in (-1, 1, 4, 6, 15, 17, 21, 34):
        for chunk in (0, 128, 256):
            cfg = apply.make_cfg(_capi.KERNEL_MARCH, v, chunk) if v >= 0 else None
            if v < 0 and chunk: continue
            try:
                ms = apply.time_builtin(body, [a], b, bounds, cfg=cfg, warmup=3, reps=12)
            except Exception as e:
                print(shape, v, chunk, "ERR", str(e)[:80]); continue
            name = lib.neptune_hip_march_variant_name(3, v).decode() if v >= 0 else "auto"
            print(f"{shape[0]}^3 {name:40s} chunk={chunk:4d} {ms:8.4f} ms {nbytes / ms / 1e6:8.1f} GB/s", flush=True)
    del a, b
    torch.cuda.empty_cache()

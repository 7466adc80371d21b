#!/usr/bin/env python3
"""Node-centred grids (2^k + 1 cells per side): rows are not a whole number of 16-byte lane vectors.  Times
the built-in bodies on such fields (march kernel with unaligned vector accesses + direct tail launch)
next to the aligned size below and to the direct kernel.   usage: tools/ragged_bench.py"""
import json
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "neptune-pde-solver_amd"))


def main():
    import torch
    from neptune_hip import _capi, apply, fields
    lib = _capi.load()
    lib.neptune_hip_init(0)
    cases = [("lap3d7_f64", (1024, 1024, 1024)), ("lap3d7_f64", (1025, 1025, 1025)), ("lap3d7_f64", (513, 513, 513)),
             ("lap2d5_f64", (8192, 8192)), ("lap2d5_f64", (8193, 8193)), ("lap3d27_f32", (513, 513, 513))]
    for body_name, shape in cases:
        body = apply.BODY_BY_NAME[body_name]
        dt = apply.BODY_DTYPE[body]
        a = fields.DeviceField.hashed(shape, dt, seed=3)
        b = fields.DeviceField.empty_like(a)
        bounds = ([1] * len(shape), [n - 1 for n in shape])
        nbytes = 2 * a.tensor.numel() * a.tensor.element_size()
        row = {"body": body_name, "shape": shape}
        for name, cfg in (("auto", None), ("direct", apply.make_cfg(_capi.KERNEL_DIRECT))):
            apply.time_builtin(body, [a], b, bounds, cfg=cfg, warmup=10, reps=5)      # clock ramp
            ms = apply.time_builtin(body, [a], b, bounds, cfg=cfg, warmup=3, reps=20)
            row[name] = {"ms": round(ms, 4), "GBps": round(nbytes / ms / 1e6, 1),
                         "kernel": lib.neptune_hip_kernel_name(apply.plan_builtin(body, [a], b, bounds, cfg=cfg)).decode()}
        print(json.dumps(row), flush=True)
        del a, b
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

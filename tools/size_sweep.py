#!/usr/bin/env python3
"""How good are the launcher's automatic choices away from the benchmark sizes?  For a spread of field sizes:
the automatic launch, the plan-time tuner's best configuration, and the direct kernel.
usage: tools/size_sweep.py"""
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
    cases = [("lap3d7_f64", (n, n, n)) for n in (96, 128, 160, 192, 256, 320, 384, 513, 640, 768)] + \
            [("lap2d5_f64", (n, n)) for n in (256, 512, 1024, 2048, 3000, 4096)] + \
            [("lap3d27_f32", (n, n, n)) for n in (128, 256, 384)]
    for body_name, shape in cases:
        body = apply.BODY_BY_NAME[body_name]
        a = fields.DeviceField.hashed(shape, apply.BODY_DTYPE[body], seed=3)
        b = fields.DeviceField.empty_like(a)
        bounds = ([1] * len(shape), [n - 1 for n in shape])
        nbytes = 2 * a.tensor.numel() * a.tensor.element_size()
        apply.time_builtin(body, [a], b, bounds, warmup=20, reps=20)
        auto = apply.time_builtin(body, [a], b, bounds, warmup=3, reps=30)
        plan = apply.plan_builtin(body, [a], b, bounds)
        cfg, best = apply.autotune_builtin(body, [a], b, bounds)
        direct = apply.time_builtin(body, [a], b, bounds, cfg=apply.make_cfg(_capi.KERNEL_DIRECT), warmup=3, reps=30)
        vname = lib.neptune_hip_march_variant_name(len(shape), int(cfg.variant)).decode() if int(cfg.variant) >= 0 else "default"
        print(json.dumps({"body": body_name, "shape": shape, "auto_kernel": lib.neptune_hip_kernel_name(plan).decode()[13:],
                          "auto_us": round(auto * 1e3, 1), "auto_GBps": round(nbytes / auto / 1e6),
                          "tuned_us": round(best * 1e3, 1), "tuned_GBps": round(nbytes / best / 1e6), "tuned": f"{vname}@{int(cfg.chunk)}",
                          "direct_us": round(direct * 1e3, 1), "direct_GBps": round(nbytes / direct / 1e6)}), flush=True)
        del a, b
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

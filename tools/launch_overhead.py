#!/usr/bin/env python3
"""What does one launch cost beyond streaming its bytes?  Time the built-in 7-point (f64) and 27-point (f32) bodies on
fields of P x 512 x 512 cells for growing P at fixed tiles / chunk lengths and fit  t(P) = t0 + P * t_plane:
t0 is the per-launch constant (grid start-up, pipeline fill, tail), 1 / t_plane the asymptotic streaming rate.
usage: tools/launch_overhead.py [--reps N]     (one MI355X; prints a table and the fit per configuration)"""
import argparse
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "neptune-pde-solver_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=40)
    args = ap.parse_args()
    import torch
    from neptune_hip import _capi, apply, fields
    lib = _capi.load()
    lib.neptune_hip_init(0)
    cases = [("3d7 f64", _capi.BODY_LAP3D7_F64, _capi.F64, 8, [(4, 64), (4, 128), (0, 64), (6, 128), (6, 256)]),
             ("3d27 f32", _capi.BODY_LAP3D27_F32, _capi.F32, 4, [(1, 64), (1, 128), (1, 32), (7, 64)])]
    planes = [128, 256, 512, 1024, 2048]
    # warm the clocks
    a = fields.DeviceField.hashed((512, 512, 512), _capi.F64, seed=1)
    b = fields.DeviceField.empty_like(a)
    apply.time_builtin(_capi.BODY_LAP3D7_F64, [a], b, ([1, 1, 1], [511, 511, 511]), warmup=50, reps=1500)
    del a, b
    for name, body, dt, esize, cfgs in cases:
        for variant, chunk in cfgs:
            ts = []
            for p in planes:
                shape = (p, 512, 512)
                fa = fields.DeviceField.hashed(shape, dt, seed=3)
                fb = fields.DeviceField.empty_like(fa)
                cfg = apply.make_cfg(_capi.KERNEL_MARCH, variant, chunk)
                ms = min(apply.time_builtin(body, [fa], fb, ([1, 1, 1], [n - 1 for n in shape]), cfg=cfg, warmup=5, reps=args.reps)
                         for _ in range(3))
                ts.append(ms)
                del fa, fb
                torch.cuda.empty_cache()
            A = np.vstack([np.ones(len(planes)), np.array(planes, float)]).T
            (t0, tp), *_ = np.linalg.lstsq(A, np.array(ts), rcond=None)
            plane_bytes = 2 * 512 * 512 * esize
            tile = lib.neptune_hip_march_variant_name(3, variant).decode()
            print(f"{name} tile {variant} {tile:28s} chunk {chunk:3d}: " + "  ".join(f"P={p}: {t * 1e3:7.1f} us" for p, t in zip(planes, ts)))
            print(f"      fit: t0 = {t0 * 1e3:6.1f} us per launch, streaming {plane_bytes / (tp * 1e-3) / 1e12:5.2f} TB/s asymptotic; "
                  f"512^3 at {512 * plane_bytes / (ts[2] * 1e-3) / 1e12:5.2f} TB/s = {512 * plane_bytes / (ts[2] * 1e-3) / 8e12 * 100:4.1f} %")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""One-off: a 2048^3 fp64 field (64 GiB, 8.6e9 cells: element indices beyond 2^32) through the march and the direct
kernel; affine integer field -> exactly 0 at every interior cell, rim = input; times both.  usage: tools/big_field_check.py [N]"""
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "neptune-pde-solver_amd"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    import torch
    from neptune_hip import _capi, apply, fields
    lib = _capi.load()
    lib.neptune_hip_init(0)
    shape = (n, n, n)
    idx = torch.arange(n, device="cuda", dtype=torch.float64)
    u = torch.zeros(shape, dtype=torch.float64, device="cuda")
    u += idx.reshape(n, 1, 1) * 3
    u += idx.reshape(1, n, 1) * 5
    u += idx.reshape(1, 1, n) * 7
    fin = fields.DeviceField((0, 0, 0), shape, _capi.F64, u)
    fout = fields.DeviceField.empty_like(fin)
    bounds = ([1, 1, 1], [n - 1] * 3)
    body = _capi.BODY_LAP3D7_F64
    for name, cfg in (("march", None), ("direct", apply.make_cfg(_capi.KERNEL_DIRECT))):
        fout.tensor.fill_(-1.0)
        apply.apply_builtin(body, [fin], fout, bounds, cfg=cfg)
        torch.cuda.synchronize()
        out = fout.tensor
        bad = 0
        for i0 in range(1, n - 1, 256):                      # slab-wise: keeps the temporaries small
            i1 = min(i0 + 256, n - 1)
            bad += int((out[i0:i1, 1:-1, 1:-1] != 0).sum())
        rim_ok = bool((out[0] == u[0]).all() and (out[-1] == u[-1]).all() and (out[:, 0] == u[:, 0]).all()
                      and (out[:, -1] == u[:, -1]).all() and (out[:, :, 0] == u[:, :, 0]).all() and (out[:, :, -1] == u[:, :, -1]).all())
        ms = apply.time_builtin(body, [fin], fout, bounds, cfg=cfg, warmup=1, reps=3)
        print(f"{name}: {n}^3 fp64 interior nonzeros={bad} rim_ok={rim_ok} {ms:.2f} ms {2 * n**3 * 8 / ms / 1e6:.0f} GB/s "
              f"kernel={lib.neptune_hip_kernel_name(apply.plan_builtin(body, [fin], fout, bounds, cfg=cfg)).decode()}", flush=True)
        assert bad == 0 and rim_ok
    print("BIG_FIELD_OK")


if __name__ == "__main__":
    main()

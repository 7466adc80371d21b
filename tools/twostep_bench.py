#!/usr/bin/env python3
"""Steps per second of an explicit time loop with one apply per pass over HBM and with two
(neptune_hip_step_loop_pairs / csrc/kernels/apply_march2.hpp): the built-in 7-point operator and a lowered module's
fused Euler step u + dt * lap(u).   usage: tools/twostep_bench.py [N ...]   (default 1024 512)"""
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "neptune-pde-solver_amd"))
sys.path.insert(0, str(REPO / "tools"))


def main():
    import torch
    import make_stencil_mlir
    from neptune_hip import _capi, apply, fields, lowering
    lib = _capi.load()
    lib.neptune_hip_init(0)
    two_d = "--2d" in sys.argv
    sizes = [int(x) for x in sys.argv[1:] if x.isdigit()] or ([8192, 2048, 1024] if two_d else [1024, 512])
    only_builtin = "--builtin-only" in sys.argv
    for n in sizes:
        rank = 2 if two_d else 3
        shape = (n,) * rank
        steps = (42 if n >= 1024 else 204) if not two_d else (204 if n >= 4096 else 1020)
        bounds = ([1] * rank, [n - 1] * rank)
        kind, opname, builtin = ("2d5", "lap2d", _capi.BODY_LAP2D5_F64) if two_d else ("3d7", "lap3d", _capi.BODY_LAP3D7_F64)
        cases = [(f"built-in {5 if two_d else 7}-point operator", builtin)]
        if not only_builtin:
            mod = lowering.compile_module(make_stencil_mlir.stencil_module(kind, list(shape), time_step=1e-3))
            cases += [(f"lowered @{opname}", mod.geom_entry(opname)), ("lowered fused Euler @step", mod.geom_entry("step"))]
        chunk = int(os.environ.get("TWOSTEP_CHUNK", "0"))
        for name, body in cases:
            a = fields.DeviceField.hashed(shape, _capi.F64, seed=5)
            a.tensor.mul_(1e-3)
            b = fields.DeviceField.empty_like(a)
            row = {"field": f"{n}^{rank} f64", "body": name, "steps": steps, "shape_variant": os.environ.get("NEPTUNE_HIP_MARCH2", "0"), "chunk": chunk}
            for label, env in (("one_apply_per_pass", "NEPTUNE_HIP_NO_PAIRS"), ("two_applies_per_pass", "NEPTUNE_HIP_NO_TRIPLES"),
                               ("three_applies_per_pass", "")):
                os.environ.pop("NEPTUNE_HIP_NO_PAIRS", None)
                os.environ.pop("NEPTUNE_HIP_NO_TRIPLES", None)
                if env:
                    os.environ[env] = "1"
                apply.step_loop(body, a, b, bounds, 60, cfg=apply.make_cfg(chunk=chunk) if chunk and env != "NEPTUNE_HIP_NO_PAIRS" else None)   # warm: graph capture, clocks
                a.fill_hash(5)
                a.tensor.mul_(1e-3)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                apply.step_loop(body, a, b, bounds, steps, cfg=apply.make_cfg(chunk=chunk) if chunk and env != "NEPTUNE_HIP_NO_PAIRS" else None)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 1e3 / steps
                row[label + "_ms_per_step"] = round(ms, 4)
                row[label + "_GBps_algorithmic"] = round(2 * n ** rank * 8 / ms / 1e6, 1)
            row["speedup"] = round(row["one_apply_per_pass_ms_per_step"] / row["two_applies_per_pass_ms_per_step"], 3)
            row["speedup3"] = round(row["one_apply_per_pass_ms_per_step"] / row["three_applies_per_pass_ms_per_step"], 3)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()

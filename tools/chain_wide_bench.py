#!/usr/bin/env python3
"""Steps per second of explicit loops beyond the 7-point family: the 13-point 4th-order operator (radius 2) and a 7-point
operator with a coefficient field read at the centre, one apply per pass against two (three) per pass
(csrc/kernels/apply_march2.hpp, round 3).   usage: tools/chain_wide_bench.py [N ...]   (default 512)"""
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "neptune-pde-solver_amd"))
sys.path.insert(0, str(REPO / "tests"))


def main():
    import torch
    import test_multihalo_gpu as mh
    from neptune_hip import _capi, apply, fields, lowering
    lib = _capi.load()
    lib.neptune_hip_init(0)
    sizes = [int(x) for x in sys.argv[1:] if x.isdigit()] or [512]
    for n in sizes:
        shape = (n, n, n)
        for name, nin, r in (("13-point (radius 2)", 1, 2), ("7-point + coefficient field", 2, 1), ("13-point + coefficient field", 2, 2)):
            acc = [(0, o) for o in mh.star(3, r)] + [(k, (0, 0, 0)) for k in range(1, nin)]
            bounds = ([r] * 3, [n - r] * 3)
            mod = lowering.compile_module(mh.module_text(shape, "f64", nin, acc, bounds[0], bounds[1]))
            entry = mod.geom_entry("resid")
            a = fields.DeviceField.hashed(shape, _capi.F64, seed=5)
            a.tensor.mul_(1e-3)
            b = fields.DeviceField.empty_like(a)
            others = [fields.DeviceField.hashed(shape, _capi.F64, seed=7 + k) for k in range(1, nin)]
            for o in others:
                o.tensor.mul_(1e-3)
            steps = 60 if n >= 1024 else 204
            row = {"field": f"{n}^3 f64", "body": name, "steps": steps, "shape": os.environ.get("NEPTUNE_HIP_MARCH2", "0")}
            for label, env in (("one", "NEPTUNE_HIP_NO_PAIRS"), ("two", "NEPTUNE_HIP_NO_TRIPLES"), ("three", "")):
                os.environ.pop("NEPTUNE_HIP_NO_PAIRS", None)
                os.environ.pop("NEPTUNE_HIP_NO_TRIPLES", None)
                if env:
                    os.environ[env] = "1"
                apply.step_loop(entry, a, b, bounds, 60, others=others)
                a.fill_hash(5)
                a.tensor.mul_(1e-3)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                apply.step_loop(entry, a, b, bounds, steps, others=others)
                torch.cuda.synchronize()
                row[label + "_ms_per_step"] = round((time.perf_counter() - t0) * 1e3 / steps, 4)
            row["speedup2"] = round(row["one_ms_per_step"] / row["two_ms_per_step"], 3)
            row["speedup3"] = round(row["one_ms_per_step"] / row["three_ms_per_step"], 3)
            print(json.dumps(row), flush=True)
            del a, b, others
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

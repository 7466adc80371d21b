#!/usr/bin/env python3
"""Time a lowered module's @entry(out, in...) on device-resident fields.
usage: tools/time_module.py <file.mlir> [--reps N]      (field shape comes from the module's types)"""
import argparse
import json
import re
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "neptune-pde-solver_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mlir")
    ap.add_argument("--symbol", default="entry")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--compile-only", action="store_true",
                    help="fill the module cache and exit without touching the GPU (run this BEFORE profiling: a cache miss "
                         "inside a rocprofv3 --pmc run would start hipcc from a profiled process)")
    args = ap.parse_args()
    from neptune_hip import lowering
    text = Path(args.mlir).read_text()
    if args.compile_only:
        lowering.compile_module(text, load=False)
        return
    import torch
    mod = lowering.compile_module(text)
    sig = mod.signatures[args.symbol]
    m = re.search(r"#b\s*=\s*#neptune_ir.bounds<lb = \[([^\]]*)\], ub = \[([^\]]*)\]>", text)
    lb = [int(x) for x in m.group(1).split(",")]
    ub = [int(x) for x in m.group(2).split(",")]
    shape = [u - l for l, u in zip(lb, ub)]
    dt = torch.float64 if sig["args"][0]["elem"] == "f64" else torch.float32
    nin = sum(1 for x in sig["args"] if x.get("kind", "memref") == "memref") - 1
    ins = [torch.rand(shape, dtype=dt, device="cuda") * 2 - 1 for _ in range(nin)]
    b = torch.zeros(shape, dtype=dt, device="cuda")
    for _ in range(3):
        mod.call(args.symbol, b, *ins)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        mod.call(args.symbol, b, *ins)       # each call synchronises at exit (ABI: results are ready)
    torch.cuda.synchronize()
    dtm = (time.perf_counter() - t0) / args.reps
    nbytes = (nin + 1) * b.numel() * b.element_size()   # every input read once, the result written once
    print(json.dumps({"module": args.mlir, "shape": shape, "ms_per_call": dtm * 1e3, "GBps": nbytes / dtm / 1e9,
                      "applies": mod.report["applies"]}))


if __name__ == "__main__":
    main()

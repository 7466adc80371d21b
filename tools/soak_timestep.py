#!/usr/bin/env python3
"""One-off soak of explicit time_advance: random single-input rhs operators (tests/test_fuzz_gpu.py generator), fused
form (rhs = one apply) and unfused form (rhs = an apply of an apply), three chained steps on device-resident fields,
every kernel form, bit for bit against the oracle.   usage: tools/soak_timestep.py FIRST_SEED COUNT"""
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "neptune-pde-solver_amd"))
sys.path.insert(0, str(REPO / "tests"))


def one_input_apply(rng, name, rank, shape, origin):
    import test_fuzz_gpu as fz
    while True:
        text, nin = fz.gen_apply(rng, name, rank, "f64", shape, origin)
        if nin == 1:
            return text


def build(seed):
    rng = np.random.default_rng(seed)
    rank = int(rng.choice([1, 2, 3, 3]))
    last = int(rng.choice([128, 192, 256, 320])) + (1 if rng.random() < 0.5 else 0)
    shape = [int(rng.integers(6, 14)) for _ in range(rank - 1)] + [last]
    origin = [int(rng.integers(-3, 5)) for _ in range(rank)]
    fused = rng.random() < 0.6
    a = one_input_apply(rng, "rhs", rank, shape, origin)
    if not fused:                                          # rhs = second(first(u)): two applies, not fusable
        b = one_input_apply(rng, "tmp", rank, shape, origin)
        la, lb = a.split("\n"), b.split("\n")
        body_b = lb[lb.index(next(l for l in lb if "%r = neptune_ir.apply(" in l)):lb.index(next(l for l in lb if l.strip() == "neptune_ir.return %r : !t"))]
        body_b[0] = body_b[0].replace("%r = neptune_ir.apply(%u0)", "%q = neptune_ir.apply(%r)")
        k = la.index(next(l for l in la if l.strip() == "neptune_ir.return %r : !t"))
        a = "\n".join(la[:k] + body_b + ["    neptune_ir.return %q : !t"] + la[k + 1:])
    mr = "x".join("?" * rank) + "xf64"
    lbs = ", ".join(map(str, origin))
    ubs = ", ".join(str(o + n) for o, n in zip(origin, shape))
    dt = float(rng.choice([0.125, 0.0625, 0.5, 0.01]))
    T = ['#l = #neptune_ir.location<"cell">', f"#b = #neptune_ir.bounds<lb = [{lbs}], ub = [{ubs}]>",
         "!t = !neptune_ir.temp<element = f64, bounds = #b, location = #l>",
         "!f = !neptune_ir.field<element = f64, bounds = #b, location = #l>", "module {", a,
         f"  func.func @step(%out: memref<{mr}>, %in: memref<{mr}>) -> memref<{mr}> {{",
         f"    %fo = neptune_ir.wrap %out : memref<{mr}> -> !f", f"    %fi = neptune_ir.wrap %in : memref<{mr}> -> !f",
         "    %u0 = neptune_ir.load %fi : !f -> !t", f"    %dt = arith.constant {dt!r} : f64",
         "    %u1 = neptune_ir.time_advance %u0, %dt {method = 0 : i32, rhs = @rhs} : !t, f64 -> !t",
         "    neptune_ir.store %u1 to %fo : !t to !f", f"    %res = neptune_ir.unwrap %fo : !f -> memref<{mr}>",
         f"    func.return %res : memref<{mr}>", "  }", "}"]
    return "\n".join(T) + "\n", tuple(shape), fused


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    os.environ["NEPTUNE_CACHE_DIR"] = tempfile.mkdtemp(prefix="neptune_soak_")
    import torch
    import helpers
    from helpers import oracle
    from neptune_hip import lowering
    t0 = time.time()
    checks = 0
    for seed in range(first, first + count):
        text, shape, fused = build(seed)
        m = oracle.Module.parse(text)
        mod = lowering.compile_module(text)
        nstep = [a["inputs"] for a in mod.report["applies"] if a["function"] == "step"]
        assert nstep == [1 if fused else 2], (nstep, fused)
        u = helpers.hash_field(shape, np.float64, seed=seed) * 0.5
        ha, hb = u.copy(), np.zeros_like(u)
        for _ in range(3):
            m.call("step", hb, ha)
            ha, hb = hb, ha
        if not np.isfinite(ha).all():
            print(f"seed {seed}: skipped (not finite after 3 steps)")
            continue
        for s in [{}, {"NEPTUNE_HIP_KERNEL": "direct"}, {"NEPTUNE_HIP_KERNEL": "direct-flat"}] + \
                 [{"NEPTUNE_HIP_VARIANT": str(v), "NEPTUNE_HIP_CHUNK": "2"} for v in range({3: 7, 2: 3, 1: 1}[len(shape)])]:
            for k in ("NEPTUNE_HIP_KERNEL", "NEPTUNE_HIP_VARIANT", "NEPTUNE_HIP_CHUNK"):
                os.environ.pop(k, None)
            os.environ.update(s)
            a, b = torch.from_numpy(u).cuda(), torch.zeros(shape, dtype=torch.float64, device="cuda")
            for _ in range(3):
                mod.call("step", b, a)
                a, b = b, a
            checks += 1
            if not helpers.bits_equal(a.cpu().numpy(), ha):
                print(f"MISMATCH seed={seed} shape={shape} fused={fused} {s}")
                print(helpers.mismatch_report(a.cpu().numpy(), ha))
                print(text)
                sys.exit(1)
        print(f"seed {seed}: shape={shape} fused={fused} ok ({checks} checks, {time.time() - t0:.0f} s)", flush=True)
    print(f"SOAK_TIMESTEP_OK seeds={first}..{first + count - 1} checks={checks} seconds={time.time() - t0:.0f}")


if __name__ == "__main__":
    main()

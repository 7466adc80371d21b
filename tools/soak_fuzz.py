#!/usr/bin/env python3
"""One-off soak of the lowering route: tests/test_fuzz_gpu.py's random modules for a range of seeds
(lower -> hipcc -> module ABI -> every kernel form), bit for bit against the oracle.
usage: tools/soak_fuzz.py FIRST_SEED COUNT"""
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "neptune-pde-solver_amd"))
sys.path.insert(0, str(REPO / "tests"))


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    os.environ["NEPTUNE_CACHE_DIR"] = tempfile.mkdtemp(prefix="neptune_soak_")
    import torch
    import helpers
    import test_fuzz_gpu as fz
    from helpers import oracle
    from neptune_hip import lowering
    t0 = time.time()
    checks = 0
    for seed in range(first, first + count):
        text, shape, elem, ops = fz.gen_module(seed)
        dt = np.float64 if elem == "f64" else np.float32
        m = oracle.Module.parse(text)
        mod = lowering.compile_module(text)
        kern = {a["function"]: a["kernel"] for a in mod.report["applies"]}
        for name, nin in ops:
            ins = [helpers.hash_field(shape, dt, seed=seed + 7 * k) for k in range(nin)]
            want = m.call(name, *ins)
            d_ins = [torch.from_numpy(a).cuda() for a in ins]
            settings = [{}, {"NEPTUNE_HIP_KERNEL": "direct"}, {"NEPTUNE_HIP_KERNEL": "direct-flat"}]
            if kern[name] == "march":
                nvar = {3: 8, 2: 3, 1: 1}[len(shape)]
                settings += [{"NEPTUNE_HIP_KERNEL": "march", "NEPTUNE_HIP_VARIANT": str(v), "NEPTUNE_HIP_CHUNK": c}
                             for v in range(nvar) for c in ("1", "5")]
            for s in settings:
                for k in ("NEPTUNE_HIP_KERNEL", "NEPTUNE_HIP_VARIANT", "NEPTUNE_HIP_CHUNK"):
                    os.environ.pop(k, None)
                os.environ.update(s)
                got = mod.call(name, *d_ins).cpu().numpy()
                checks += 1
                if not helpers.bits_equal(got, want):
                    print(f"MISMATCH seed={seed} {name} shape={shape} {elem} kernel={kern[name]} {s}")
                    print(helpers.mismatch_report(got, want))
                    print(text)
                    sys.exit(1)
        for k in ("NEPTUNE_HIP_KERNEL", "NEPTUNE_HIP_VARIANT", "NEPTUNE_HIP_CHUNK"):
            os.environ.pop(k, None)
        nmax = max(n for _, n in ops)
        ins = [helpers.hash_field(shape, dt, seed=seed + 11 * k) for k in range(nmax)]
        for inplace in (False, True):                # the composed @entry, fresh destination and in place
            h_ins = [a.copy() for a in ins]
            h_out = h_ins[0] if inplace else np.full(shape, 9.0, dtype=dt)
            m.call("entry", h_out, *h_ins)
            g_ins = [torch.from_numpy(a.copy()).cuda() for a in ins]
            g_out = g_ins[0] if inplace else torch.full(shape, 9.0, dtype=g_ins[0].dtype, device="cuda")
            mod.call("entry", g_out, *g_ins)
            checks += 1
            if not helpers.bits_equal(g_out.cpu().numpy(), h_out):
                print(f"MISMATCH seed={seed} entry inplace={inplace} shape={shape} {elem}")
                print(helpers.mismatch_report(g_out.cpu().numpy(), h_out))
                print(text)
                sys.exit(1)
        print(f"seed {seed}: shape={shape} {elem} ok ({checks} checks, {time.time() - t0:.0f} s)", flush=True)
    print(f"SOAK_FUZZ_OK seeds={first}..{first + count - 1} checks={checks} seconds={time.time() - t0:.0f}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Inputs in boxes of their own at bench size: a 512^3 fp64 cell-located result from a cell field (centre), a field on the
k-faces (extent N+1, read at k and k+1) and a field that carries two ghost layers (radius-2 star, or 7-point) -- the
staggered-grid shape tests/test_ownbox_gpu.py checks bit for bit.  Times @entry on the automatic kernel and on the direct
kernel (where such applies ran before round 3).  Bytes counted: every input cell read once + the result written once.
usage: tools/ownbox_bench.py [N]"""
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "neptune-pde-solver_amd"))
sys.path.insert(0, str(REPO / "tests"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    import torch
    import test_ownbox_gpu as ob
    from neptune_hip import lowering
    box = ob.box((0, 0, 0), (n, n, n))
    cases = {
        "faces_k + ghosts2 radius-2 star": ([box, ob.grow(box, (0, 0, 0), (0, 0, 1)), ob.grow(box, (2, 2, 2), (2, 2, 2))],
                                            [(0, (0, 0, 0)), (1, (0, 0, 0)), (1, (0, 0, 1)), (2, (0, 0, 0))] + [(2, o) for o in ob.star(3, 2)]),
        "faces_k + ghosts2 7-point": ([box, ob.grow(box, (0, 0, 0), (0, 0, 1)), ob.grow(box, (2, 2, 2), (2, 2, 2))],
                                      [(0, (0, 0, 0)), (1, (0, 0, 0)), (1, (0, 0, 1)), (2, (0, 0, 0))] + [(2, o) for o in ob.star(3, 1)]),
        "ghosted 7-point (two fields)": ([box, ob.grow(box, (1, 1, 1), (1, 1, 1))], [(0, (0, 0, 0)), (1, (0, 0, 0))] + [(1, o) for o in ob.star(3, 1)]),
        "faces_i + faces_j + faces_k": ([box, ob.grow(box, (0, 0, 0), (1, 0, 0)), ob.grow(box, (0, 0, 0), (0, 1, 0)), ob.grow(box, (0, 0, 0), (0, 0, 1))],
                                        [(0, (0, 0, 0)), (1, (0, 0, 0)), (1, (1, 0, 0)), (2, (0, 0, 0)), (2, (0, 1, 0)), (3, (0, 0, 0)), (3, (0, 0, 1))]),
    }
    for name, (in_boxes, accesses) in cases.items():
        text = ob.module_text("f64", box, box, in_boxes, accesses)
        mod = lowering.compile_module(text)
        ins = [torch.rand(tuple(u - l for l, u in zip(*b)), dtype=torch.float64, device="cuda") for b in in_boxes]
        out = torch.zeros((n, n, n), dtype=torch.float64, device="cuda")
        nbytes = sum(t.numel() for t in ins) * 8 + out.numel() * 8
        row = []
        for kern in ("auto", "direct"):
            os.environ.pop("NEPTUNE_HIP_KERNEL", None)
            if kern == "direct":
                os.environ["NEPTUNE_HIP_KERNEL"] = "direct"
            for _ in range(5):
                mod.call("entry", out, *ins)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 30
            for _ in range(reps):
                mod.call("entry", out, *ins)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            row.append(f"{kern}: {ms:.3f} ms = {nbytes / ms / 1e9:.2f} TB/s ({nbytes / ms / 1e9 / 8 * 100:.0f} %)")
        os.environ.pop("NEPTUNE_HIP_KERNEL", None)
        print(f"{name:34s} {n}^3 f64, {len(ins)} inputs: " + "   ".join(row), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Emit one of the several-halo-input test modules (tests/test_multihalo_gpu.py CASES) at any size.
usage: make_multihalo_mlir.py CASE N0 [N1 [N2]] > out.mlir"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))


def main():
    import test_multihalo_gpu as mh
    name = sys.argv[1]
    _, elem, nin, accesses, margin, _ = mh.CASES[name]
    shape = [int(x) for x in sys.argv[2:]]
    sys.stdout.write(mh.module_text(shape, elem, nin, accesses, [margin] * len(shape), [n - margin for n in shape]))


if __name__ == "__main__":
    main()

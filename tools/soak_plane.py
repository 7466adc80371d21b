#!/usr/bin/env python3
"""One-off soak of the plane-in-LDS kernels (csrc/kernels/apply_plane.hpp; every fifth seed: the rank-2 LDS tile kernel): random 3-D stars of radius 2..8 (unequal radii
per axis, optional second input read at the centre) and random radius-2 boxes, random field shapes (windows and chunks
cut inside the field, ragged rows), tight bounds; every default tile x chunk seams + the direct kernel, bit for bit against
the oracle.  Modules are compiled by a pool of host threads.      usage: tools/soak_plane.py FIRST_SEED COUNT [THREADS]"""
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "neptune-pde-solver_amd"))
sys.path.insert(0, str(REPO / "tests"))


def gen_case(seed):
    import test_multihalo_gpu as mh
    rng = np.random.default_rng(seed)
    elem = str(rng.choice(["f64", "f64", "f32"]))
    vk = 2 if elem == "f64" else 4
    if seed % 5 == 4:
        return gen_case_2d(rng, elem, vk)
    box = rng.random() < 0.25
    if box:
        rad = [int(rng.integers(1, 3)), int(rng.integers(0, 3)), int(rng.integers(0, 3))]
        rad[int(rng.integers(1, 3))] = 2 if max(rad) < 2 else rad[int(rng.integers(1, 3))]
        if max(rad) < 2:
            rad[0] = 2
        taps = {(0, 0, 0)}
        for _ in range(int(rng.integers(6, 40))):
            taps.add(tuple(int(rng.integers(-r, r + 1)) for r in rad))
        taps.add(tuple(int(rng.choice([-r, r])) if r else 0 for r in rad))      # one corner: makes it a box for sure
        accesses = [(0, o) for o in sorted(taps)]
        nin = 1
    else:
        rad = [int(rng.integers(0, 9)), int(rng.integers(0, 9)), int(rng.integers(0, 9))]
        if max(rad) < 2:
            rad[int(rng.integers(0, 3))] = int(rng.integers(2, 9))
        if rad[1] == 0 and rad[2] == 0:
            rad[int(rng.integers(1, 3))] = int(rng.integers(1, 5))
        taps = [(0, 0, 0)]
        for d in range(3):
            for sgn in (-1, 1):
                for dist in range(1, rad[d] + 1):
                    if dist == rad[d] or rng.random() < 0.7:
                        o = [0, 0, 0]
                        o[d] = sgn * dist
                        taps.append(tuple(o))
        nin = 2 if rng.random() < 0.3 else 1
        halo = int(rng.integers(0, nin))
        accesses = [(halo, o) for o in taps]
        if nin == 2:
            accesses.insert(int(rng.integers(0, len(accesses))), (1 - halo, (0, 0, 0)))
        if halo != 0 and (0, (0, 0, 0)) not in accesses:
            accesses.insert(0, (0, (0, 0, 0)))
    accesses = accesses[:len(mh.COEF)]
    n0 = 2 * rad[0] + int(rng.integers(2, 12))
    n1 = 2 * rad[1] + int(rng.choice([3, 7, 20, 33, 45, 70]))
    n2 = int(rng.choice([128, 192, 256, 320, 520])) * (vk // 2) + (int(rng.integers(1, vk)) if rng.random() < 0.4 else 0)
    n2 = max(n2, 64 * vk + 2 * rad[2] + 2 * vk)
    shape = (n0, n1, n2)
    lb = [rad[d] + int(rng.integers(0, 2)) for d in range(3)]
    ub = [shape[d] - rad[d] - int(rng.integers(0, 2)) for d in range(3)]
    text = mh.module_text(shape, elem, nin, accesses, lb, ub)
    return text, shape, elem, nin, rad, box


def gen_case_2d(rng, elem, vk):
    """rank 2, beyond the march kernel's registers: the LDS tile kernel (neptune_apply_tile2)"""
    import test_multihalo_gpu as mh
    kind = str(rng.choice(["star", "box", "multi"]))
    nin = 1
    if kind == "star":
        rad = [int(rng.integers(5, 9)), int(rng.integers(0, 9))] if rng.random() < 0.5 else [int(rng.integers(0, 9)), int(rng.integers(5, 9))]
        taps = [(0, 0)] + [tuple(sg * dist if a == d else 0 for a in range(2)) for d in range(2) for sg in (-1, 1)
                           for dist in range(1, rad[d] + 1) if dist == rad[d] or rng.random() < 0.7]
        accesses = [(0, o) for o in taps]
    elif kind == "box":
        rad = [int(rng.integers(1, 5)), int(rng.integers(1, 5))]
        rad[int(rng.integers(0, 2))] = int(rng.integers(3, 5))
        taps = {(0, 0), (rad[0], -rad[1]), (-rad[0], rad[1])}
        for _ in range(int(rng.integers(6, 50))):
            taps.add((int(rng.integers(-rad[0], rad[0] + 1)), int(rng.integers(-rad[1], rad[1] + 1))))
        accesses = [(0, o) for o in sorted(taps)]
    else:
        nin = int(rng.integers(2, 5))
        rad = [int(rng.integers(3, 6)), int(rng.integers(1, 6))]
        accesses = [(0, (0, 0))]
        for k in range(nin):
            for d in range(2):
                for sg in (-1, 1):
                    accesses.append((k, tuple(sg * rad[d] if a == d else 0 for a in range(2))))
                    if rad[d] > 1 and rng.random() < 0.5:
                        accesses.append((k, tuple(sg * int(rng.integers(1, rad[d])) if a == d else 0 for a in range(2))))
        accesses = list(dict.fromkeys(accesses))
    accesses = accesses[:len(mh.COEF)]
    n0 = 2 * rad[0] + int(rng.choice([3, 9, 30, 41, 70]))
    n1 = int(rng.choice([128, 192, 256, 320, 520])) * (vk // 2) + (int(rng.integers(1, vk)) if rng.random() < 0.4 else 0)
    n1 = max(n1, 64 * vk + 2 * rad[1] + 2 * vk)
    shape = (n0, n1)
    lb = [rad[d] + int(rng.integers(0, 2)) for d in range(2)]
    ub = [shape[d] - rad[d] - int(rng.integers(0, 2)) for d in range(2)]
    return mh.module_text(shape, elem, nin, accesses, lb, ub), shape, elem, nin, rad, kind != "star"


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    os.environ["NEPTUNE_CACHE_DIR"] = tempfile.mkdtemp(prefix="neptune_soak_plane_")
    import torch
    import helpers
    from helpers import oracle
    from neptune_hip import lowering
    cases = {seed: gen_case(seed) for seed in range(first, first + count)}
    t0 = time.time()
    checks = launches = 0
    kernels = {}
    with ThreadPoolExecutor(threads) as pool:
        futs = {seed: pool.submit(lowering.compile_module, c[0]) for seed, c in cases.items()}
        for seed, (text, shape, elem, nin, rad, box) in cases.items():
            dt = np.float64 if elem == "f64" else np.float32
            mod = futs[seed].result()
            kern = mod.report["applies"][0]["kernel"]
            kernels[kern] = kernels.get(kern, 0) + 1
            ins = [helpers.hash_field(shape, dt, seed=seed + 7 * k) for k in range(nin)]
            want = np.full(shape, -7.0, dtype=dt)
            oracle.Module.parse(text).call("entry", want, *ins)
            d_ins = [torch.from_numpy(a).cuda() for a in ins]
            settings = [{}, {"NEPTUNE_HIP_KERNEL": "direct"}]
            if kern == "march":
                nvar = 8 if len(shape) == 3 else 3
                settings += [{"NEPTUNE_HIP_VARIANT": str(v), "NEPTUNE_HIP_CHUNK": c} for v in range(nvar) for c in ("1", "3", "0")]
            for s in settings:
                for k in ("NEPTUNE_HIP_KERNEL", "NEPTUNE_HIP_VARIANT", "NEPTUNE_HIP_CHUNK"):
                    os.environ.pop(k, None)
                os.environ.update(s)
                d_out = torch.full(shape, -7.0, dtype=torch.float64 if elem == "f64" else torch.float32, device="cuda")
                mod.call("entry", d_out, *d_ins)
                got = d_out.cpu().numpy()
                launches += 1
                if not np.array_equal(got.view(np.uint64 if elem == "f64" else np.uint32), want.view(np.uint64 if elem == "f64" else np.uint32)):
                    bad = np.argwhere(got != want)
                    print(f"MISMATCH seed={seed} shape={shape} {elem} rad={rad} box={box} nin={nin} {s}: {len(bad)} cells, first {bad[0].tolist()}", flush=True)
                    sys.exit(1)
            checks += 1
            print(f"seed {seed}: shape={shape} {elem} rad={rad} box={box} nin={nin} kernel={kern} ok ({time.time() - t0:.0f} s)", flush=True)
    print(f"{checks} modules, {launches} launches, all bit-exact; kernels {kernels}; {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()

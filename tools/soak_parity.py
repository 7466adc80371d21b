#!/usr/bin/env python3
"""One-off soak: many random geometries (shapes incl. ragged rows, logical origins, apply bounds, launch regions,
chunk lengths) on every march tile and both direct forms of the built-in bodies, bit for bit against the oracle.
Not part of the test suite (minutes of GPU time); usage: tools/soak_parity.py [cases] [seed]"""
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "neptune-pde-solver_amd"))
sys.path.insert(0, str(REPO / "tests"))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import torch
    import helpers
    from neptune_hip import _capi, apply, fields
    lib = _capi.load()
    lib.neptune_hip_init(0)
    rng = np.random.default_rng(seed)
    kinds = {"2d5": ("lap2d5_f64", np.float64, 2), "3d7": ("lap3d7_f64", np.float64, 3), "3d27": ("lap3d27_f32", np.float32, 3)}
    t0 = time.time()
    launches = 0
    for c in range(cases):
        kind = str(rng.choice(list(kinds)))
        body_name, dt, rank = kinds[kind]
        body = apply.BODY_BY_NAME[body_name]
        vk = 16 // np.dtype(dt).itemsize
        shape = [int(rng.integers(3, 40)) for _ in range(rank - 1)]
        if rank == 3 and rng.random() < 0.3:     # a row or two past the last whole row tile (2^k+1 rows): the row-tail launch
            shape[1] = int(rng.choice([33, 34, 65, 66, 68, 129, 130]))
        last = int(rng.choice([vk * 3, 30, 64, 126, 128, 129, 130, 131, 200, 255, 256, 257, 300, 384, 385, 513, 640]))
        shape.append(max(last, vk * 3))
        origin = [int(rng.integers(-9, 10)) for _ in range(rank)]
        lb, ub = [], []
        for d in range(rank):
            lo = int(rng.integers(1, max(2, shape[d] // 2)))
            hi = int(rng.integers(lo, shape[d]))
            lb.append(origin[d] + lo)
            ub.append(origin[d] + max(min(hi, shape[d] - 1), lo))
        u = helpers.hash_field(tuple(shape), dt, seed=int(rng.integers(1, 1 << 30)))
        want = helpers.oracle_entry(kind, u, origin, (lb, ub))
        fin = fields.DeviceField.from_numpy(u, origin)
        # a launch region: whole field, or a plane range (what slab edges / interiors are)
        regions = [None]
        if shape[0] > 4:
            a = int(rng.integers(0, shape[0] - 1))
            b = int(rng.integers(a + 1, shape[0] + 1))
            regions.append(([a] + [0] * (rank - 1), [b] + shape[1:]))
        cfgs = [apply.make_cfg(_capi.KERNEL_DIRECT), apply.make_cfg(_capi.KERNEL_DIRECT, flags=_capi.FLAG_DIRECT_FLAT), None]
        for v in range(lib.neptune_hip_march_variant_count(rank)):
            cfgs.append(apply.make_cfg(_capi.KERNEL_MARCH, v, int(rng.choice([0, 1, 2, 3, 5, 8, 16]))))
        for region in regions:
            for cfg in cfgs:
                fout = fields.DeviceField.empty_like(fin)
                fout.tensor.fill_(-3.0)
                apply.apply_builtin(body, [fin], fout, (lb, ub), region=region, cfg=cfg)
                torch.cuda.synchronize()
                got = fout.numpy()
                launches += 1
                exp = want
                if region is not None:
                    exp = np.full_like(want, -3.0)
                    exp[region[0][0]:region[1][0]] = want[region[0][0]:region[1][0]]
                if not helpers.bits_equal(got, exp):
                    print(f"MISMATCH case={c} kind={kind} shape={shape} origin={origin} bounds={(lb, ub)} region={region} "
                          f"cfg={(cfg.kernel, cfg.variant, cfg.chunk, cfg.flags) if cfg else None}")
                    print(helpers.mismatch_report(got, exp))
                    sys.exit(1)
        if c % 20 == 0:
            print(f"case {c}: {launches} launches ok, {time.time() - t0:.0f} s", flush=True)
    print(f"SOAK_OK cases={cases} seed={seed} launches={launches} seconds={time.time() - t0:.0f}")


if __name__ == "__main__":
    main()

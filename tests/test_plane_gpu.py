"""Random footprints on the LDS kernels (csrc/kernels/apply_plane.hpp), a small fixed budget of tools/soak_plane.py's
generator: 3-D stars of radius 2..8 with unequal radii per axis, a second input at the centre, radius-2 boxes, and (every
fifth seed) rank-2 stars / boxes / several wide halo inputs on the LDS tile kernel -- lowered, compiled, run on the automatic
tile, on two tile indices with chunk seams and on the direct kernel, bit for bit against the oracle.  The soak itself
(hundreds of modules) is recorded in profiles/r02_soak.txt."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

import helpers
from helpers import bits_equal, mismatch_report, oracle

pytestmark = pytest.mark.gpu
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))


@pytest.fixture(scope="module")
def env(built_libs, tmp_path_factory):
    import torch
    assert torch.cuda.is_available()
    os.environ["NEPTUNE_CACHE_DIR"] = str(tmp_path_factory.mktemp("neptune_cache_plane"))
    from neptune_hip import lowering
    import soak_plane
    helpers.prefetch_modules([soak_plane.gen_case(seed)[0] for seed in SEEDS])   # compiled side by side
    return lowering, torch


SEEDS = [3, 14, 40, 1007, 1009, 1012, 1016, 3004, 3019]


@pytest.mark.parametrize("seed", SEEDS)
def test_random_plane_footprints_match_the_oracle(env, seed):
    import soak_plane
    lowering, torch = env
    text, shape, elem, nin, rad, box = soak_plane.gen_case(seed)
    dt = np.float64 if elem == "f64" else np.float32
    mod = lowering.compile_module(text)
    assert mod.report["applies"][0]["kernel"] == "march", (shape, rad, box)
    ins = [helpers.hash_field(shape, dt, seed=seed + 7 * k) for k in range(nin)]
    want = np.full(shape, -7.0, dtype=dt)
    oracle.Module.parse(text).call("entry", want, *ins)
    d_ins = [torch.from_numpy(a).cuda() for a in ins]
    saved = {k: os.environ.get(k) for k in ("NEPTUNE_HIP_KERNEL", "NEPTUNE_HIP_VARIANT", "NEPTUNE_HIP_CHUNK")}
    try:
        settings = [{}, {"NEPTUNE_HIP_KERNEL": "direct"}]
        settings += [{"NEPTUNE_HIP_VARIANT": v, "NEPTUNE_HIP_CHUNK": c} for v in ("1", "2" if len(shape) == 2 else "7") for c in ("1", "3")]
        for s in settings:
            for k in saved:
                os.environ.pop(k, None)
            os.environ.update(s)
            d_out = torch.full(shape, -7.0, dtype=torch.float64 if elem == "f64" else torch.float32, device="cuda")
            mod.call("entry", d_out, *d_ins)
            got = d_out.cpu().numpy()
            assert bits_equal(got, want), f"seed={seed} shape={shape} {elem} rad={rad} box={box} {s}: " + mismatch_report(got, want)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

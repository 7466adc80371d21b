"""Rank body of tests/test_slab_gpu.py: several ranks share ONE GPU (gloo transport, host-staged
halos) and run the real ShardedApply + HIP kernels; every rank then checks its slab against the
single-process result of the same chained applies."""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "neptune-pde-solver_amd"))
sys.path.insert(0, str(REPO / "tests"))

import helpers  # noqa: E402
from neptune_hip import _capi, apply, fields, slab as slab_mod  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    lib = _capi.load()
    lib.neptune_hip_init(0)
    kind = os.environ.get("SLAB_KIND", "3d7")
    shape = tuple(int(x) for x in os.environ.get("SLAB_SHAPE", "24,12,256").split(","))
    steps = int(os.environ.get("SLAB_STEPS", "4"))
    body = {"3d7": _capi.BODY_LAP3D7_F64, "2d5": _capi.BODY_LAP2D5_F64, "3d27": _capi.BODY_LAP3D27_F32}[kind]
    dtype = apply.BODY_DTYPE[body]
    npdt = np.float32 if dtype == _capi.F32 else np.float64
    gbox = ([0] * len(shape), list(shape))
    gbounds = ([1] * len(shape), [n - 1 for n in shape])
    u = helpers.hash_field(shape, npdt, seed=5)

    # single-process reference on the same GPU, same kernels, whole field
    a = fields.DeviceField.from_numpy(u)
    b = fields.DeviceField.empty_like(a)
    for _ in range(steps):
        apply.apply_builtin(body, [a], b, gbounds)
        a, b = b, a
    torch.cuda.synchronize()
    want = a.numpy()
    if rank == 0:   # and that reference equals the oracle's chained applies
        o = u
        for _ in range(steps):
            o = helpers.oracle_entry(kind, o)
        assert helpers.bits_equal(want, o), "single-GPU chain differs from the oracle"

    sl = slab_mod.decompose(gbox, 1, rank, world)
    lo, hi = sl.owned_planes()
    local = np.full(sl.local_shape, np.nan, npdt)      # ghosts poisoned: the exchange must fill them
    local[lo:hi] = u[sl.start:sl.stop]
    bufs = [fields.DeviceField.from_numpy(local, sl.local_lb), fields.DeviceField(sl.local_lb, sl.local_ub, dtype)]
    bufs[1].tensor.fill_(float("nan"))
    op = slab_mod.ShardedApply(sl, body, gbounds)
    for s in range(steps):
        op(bufs[s % 2], bufs[(s + 1) % 2])
    torch.cuda.synchronize()
    got = bufs[steps % 2].numpy()[lo:hi]
    ok = helpers.bits_equal(got, want[sl.start:sl.stop])
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    if rank == 0:
        assert all(flags), f"per-rank parity: {flags}"
        print(f"SLAB_GPU_OK world={world} kind={kind} shape={shape} steps={steps}")
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        sys.exit(3)


if __name__ == "__main__":
    main()

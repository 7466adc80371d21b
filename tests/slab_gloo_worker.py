"""Rank body of tests/test_slab_gloo.py: launched by torch.distributed.run with the gloo backend.

Exercises the product's slab geometry + halo exchange (neptune_hip.slab) on CPU tensors; the
cell updates are done by the oracle standing in for the HIP kernels (this is a test: the
product path itself never touches oracle/)."""
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
from neptune_hip import slab as slab_mod  # noqa: E402


def sharded_oracle_apply(kind, local: np.ndarray, sl, global_bounds, steps_done=0):
    """one sharded apply on this rank's local buffer (ghosts already exchanged): the oracle
    evaluates exactly the geometry ShardedApply hands to the HIP kernels"""
    lb, ub = sl.clip_bounds(global_bounds)
    res = helpers.oracle_entry(kind, local, origin=list(sl.local_lb), bounds=(lb, ub))
    lo, hi = sl.owned_planes()
    out = np.array(local, copy=True)   # cells outside the launch regions keep their old value
    interior, edges = sl.regions()
    for reg in ([interior] if interior else []) + edges:
        a, b = reg[0][0], reg[1][0]
        assert lo <= a <= b <= hi
        out[a:b] = res[a:b]
    return out


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    kind = os.environ.get("SLAB_KIND", "3d7")
    shape = tuple(int(x) for x in os.environ.get("SLAB_SHAPE", "13,6,8").split(","))
    steps = int(os.environ.get("SLAB_STEPS", "3"))
    dtype = np.float32 if kind == "3d27" else np.float64
    u = helpers.hash_field(shape, dtype, seed=31)
    gbox = ([0] * len(shape), list(shape))
    gbounds = ([1] * len(shape), [n - 1 for n in shape])

    # reference result: `steps` chained whole-field applies on one process
    want = u
    for _ in range(steps):
        want = helpers.oracle_entry(kind, want)

    sl = slab_mod.decompose(gbox, 1, rank, world)
    # local buffers: owned planes from the global field, ghost planes poisoned until exchanged
    cur = np.full(sl.local_shape, np.nan, dtype)
    lo, hi = sl.owned_planes()
    cur[lo:hi] = u[sl.start:sl.stop]
    for s in range(steps):
        t = torch.from_numpy(cur)
        for w in slab_mod.exchange_halos(sl, t):
            w.wait()
        assert not np.isnan(cur).any(), "ghost planes not filled"
        nxt = sharded_oracle_apply(kind, cur, sl, gbounds)
        # ghosts of the result are stale by construction; poison them so the next exchange must refill
        if sl.r_lo:
            nxt[:lo] = np.nan
        if sl.r_hi:
            nxt[hi:] = np.nan
        cur = nxt
    got = cur[lo:hi]
    ok = helpers.bits_equal(got, want[sl.start:sl.stop])
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    # coverage: the slabs tile the global planes exactly once
    spans = [None] * world
    dist.all_gather_object(spans, (sl.start, sl.stop))
    if rank == 0:
        assert spans[0][0] == 0 and spans[-1][1] == shape[0]
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        assert all(flags), f"per-rank parity: {flags}"
        print(f"SLAB_OK world={world} kind={kind} shape={shape} steps={steps}")
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        sys.exit(3)


if __name__ == "__main__":
    main()

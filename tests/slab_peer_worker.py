"""Rank body of tests/test_slab_peer_gpu.py: two or three PROCESSES share one GPU and run the C-ABI sharded step
(neptune_hip_slab_apply) on the peer-copy transport -- IPC mappings of each other's buffers, handshake kernels, pushes,
communication stream, events, interior beside the exchange -- for real, between processes.  gloo only carries the
communicator id and the verdict.  Every rank checks its slab against the oracle's chained applies on the global field."""
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
    overlap = os.environ.get("SLAB_OVERLAP", "1") == "1"
    comm = slab_mod.SlabComm.from_process_group(transport="peer")
    gbox = ([0] * len(shape), list(shape))
    if kind == "13pt":     # a lowered module's geometry-level entry, radius 2: two ghost planes per side
        from neptune_hip import lowering
        os.environ["NEPTUNE_CACHE_DIR"] = os.environ["SLAB_CACHE_DIR"]
        text = (helpers.FIXTURE_DIR / "apply-3d-13pt.mlir").read_text()   # 20 x 18 x 256, bounds [2, 18) x [2, 16) x [2, 254)
        assert shape == (20, 18, 256)
        if rank == 0:
            lowering.compile_module(text, load=False)
        dist.barrier()
        mod = lowering.compile_module(text)
        body = mod.geom_entry("lap13")
        radius, dtype, npdt = body.halo0, _capi.F64, np.float64
        gbounds = ([2, 2, 2], [18, 16, 254])
        omod = helpers.oracle.Module.parse(text)
        oracle_step = lambda a: omod.call("lap13", a)   # noqa: E731
    else:
        body = {"3d7": _capi.BODY_LAP3D7_F64, "2d5": _capi.BODY_LAP2D5_F64, "3d27": _capi.BODY_LAP3D27_F32}[kind]
        radius, dtype = 1, apply.BODY_DTYPE[body]
        npdt = np.float32 if dtype == _capi.F32 else np.float64
        gbounds = ([1] * len(shape), [n - 1 for n in shape])
        oracle_step = lambda a: helpers.oracle_entry(kind, a)   # noqa: E731
    u = helpers.hash_field(shape, npdt, seed=5)
    want = u
    for _ in range(steps):
        want = oracle_step(want)

    sl = slab_mod.decompose(gbox, radius, rank, world)
    lo, hi = sl.owned_planes()
    local = np.full(sl.local_shape, np.nan, npdt)      # ghosts poisoned: only the exchange can fill them
    local[lo:hi] = u[sl.start:sl.stop]
    bufs = [fields.DeviceField.from_numpy(local, sl.local_lb), fields.DeviceField(sl.local_lb, sl.local_ub, dtype)]
    bufs[1].tensor.fill_(float("nan"))
    op = slab_mod.ShardedApply(sl, body, gbounds, comm=comm, overlap=overlap)
    for s in range(steps):
        op(bufs[s % 2], bufs[(s + 1) % 2])
    torch.cuda.synchronize()
    comm.status()
    got = bufs[steps % 2].numpy()[lo:hi]
    ok = helpers.bits_equal(got, want[sl.start:sl.stop])
    if not ok:
        print(f"rank {rank}: " + helpers.mismatch_report(got, want[sl.start:sl.stop]), file=sys.stderr, flush=True)
    # the timing read-out works between processes too
    op.timing(True)
    op(bufs[steps % 2], bufs[(steps + 1) % 2])
    t = op.read_timing()
    ok_t = (sl.r_lo == 0 and sl.r_hi == 0) or (t is not None and t["steps"] == 1 and t["step_ms"] > 0)
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok and ok_t))
    dist.barrier()
    del op
    torch.cuda.synchronize()
    comm.close()
    if rank == 0:
        assert all(flags), f"per-rank parity: {flags}"
        print(f"SLAB_PEER_OK world={world} kind={kind} shape={shape} steps={steps}")
    dist.destroy_process_group()
    if not (ok and ok_t):
        sys.exit(3)


if __name__ == "__main__":
    main()

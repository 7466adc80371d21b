"""Slab decomposition of one field across the GPUs of a node + halo exchange over RCCL.

The reference has no domain decomposition at all (every PETSc object is on PETSC_COMM_SELF,
lib/Runtime/PETSc/NeptunePETScRuntime.cpp:136,244,257); this is the multi-GPU face of the
`neptune_ir.apply` hot path.  Cell updates are independent given a halo of r = max |offset|
cells, so a field is cut into slabs along dim 0 (the slowest dim: a halo is then r contiguous
planes, one contiguous run of memory, no packing):

    rank g owns global planes [start_g, stop_g); its local buffer also holds r ghost planes
    from each existing neighbour:   local box = [start_g - r_lo, stop_g + r_hi) x full x full
    (r_lo = r if g > 0 else 0, r_hi = r if g < G-1 else 0 -- no ghosts beyond the global
    boundary, where the apply's copy-through semantics hold instead of a periodic wrap,
    DataflowLowering.cpp:382-410 has none).

One sharded apply (one process per GPU, torch.distributed backend "nccl" = RCCL over xGMI):

    comm stream   : send first/last r owned planes of the INPUT to the neighbours,
                    receive their planes into the ghost planes          (grouped isend/irecv)
    compute stream: interior planes (those whose neighbourhood is already local)   -- overlaps
    compute stream: after the exchange, the r edge planes on each side that needed ghosts

Everything here is geometry + torch.distributed plumbing; the cell updates themselves run in
libneptune_hip.so (GPU) -- or, in the CPU tests of this module, in the oracle.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

Box = Tuple[Sequence[int], Sequence[int]]


@dataclass(frozen=True)
class Slab:
    """what rank `rank` of `world` holds of a global field box [glb, gub)"""
    rank: int
    world: int
    radius: int
    glb: Tuple[int, ...]
    gub: Tuple[int, ...]
    start: int   # first owned global plane (logical coordinate along dim 0)
    stop: int    # one past the last owned plane
    r_lo: int    # ghost planes below / above
    r_hi: int

    # ---- boxes ---------------------------------------------------------------------------
    @property
    def local_lb(self) -> Tuple[int, ...]:
        return (self.start - self.r_lo,) + tuple(self.glb[1:])

    @property
    def local_ub(self) -> Tuple[int, ...]:
        return (self.stop + self.r_hi,) + tuple(self.gub[1:])

    @property
    def local_shape(self) -> Tuple[int, ...]:
        return tuple(u - l for l, u in zip(self.local_lb, self.local_ub))

    @property
    def n_own(self) -> int:
        return self.stop - self.start

    def owned_planes(self) -> Tuple[int, int]:
        """physical plane range of the owned part inside the local buffer"""
        return self.r_lo, self.r_lo + self.n_own

    def clip_bounds(self, bounds: Box) -> Box:
        """apply.bounds restricted to the planes this rank owns"""
        lb, ub = list(bounds[0]), list(bounds[1])
        lb[0] = max(lb[0], self.start)
        ub[0] = min(ub[0], self.stop)
        if ub[0] < lb[0]:
            ub[0] = lb[0]
        return lb, ub

    def regions(self) -> Tuple[Optional[Box], List[Box]]:
        """(interior, edges) as result-physical boxes of the local buffer.  Interior planes
        need no ghost data; edge planes (r on each side that has a neighbour) do."""
        shape = self.local_shape
        lo, hi = self.owned_planes()
        i_lo = lo + (self.radius if self.r_lo else 0)
        i_hi = hi - (self.radius if self.r_hi else 0)
        full = lambda a, b: ([a] + [0] * (len(shape) - 1), [b] + list(shape[1:]))
        if i_hi <= i_lo:  # slab thinner than two halos: everything waits for the exchange
            return None, [full(lo, hi)]
        edges = []
        if self.r_lo:
            edges.append(full(lo, i_lo))
        if self.r_hi:
            edges.append(full(i_hi, hi))
        return full(i_lo, i_hi), edges


def decompose(global_box: Box, radius: int, rank: int, world: int) -> Slab:
    glb, gub = tuple(int(x) for x in global_box[0]), tuple(int(x) for x in global_box[1])
    n0 = gub[0] - glb[0]
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    if n0 < world:
        raise ValueError(f"cannot cut {n0} planes into {world} slabs")
    base, rem = divmod(n0, world)
    start = glb[0] + rank * base + min(rank, rem)
    stop = start + base + (1 if rank < rem else 0)
    if world > 1 and base < radius:
        raise ValueError("slab thinner than the halo radius")
    return Slab(rank, world, radius, glb, gub, start, stop, radius if rank > 0 else 0,
                radius if rank < world - 1 else 0)


def halo_ops(slab: Slab, t: torch.Tensor, group=None) -> List[dist.P2POp]:
    """isend/irecv descriptors that fill the ghost planes of local buffer `t` (any device):
    planes are contiguous dim-0 slices, sent and received in place."""
    if tuple(t.shape) != slab.local_shape or not t.is_contiguous():
        raise ValueError("tensor is not this slab's dense local buffer")
    r = slab.radius
    lo, hi = slab.owned_planes()
    ops: List[dist.P2POp] = []
    if slab.r_lo:  # neighbour below (rank-1): my first r owned planes go down, its last r come up
        ops.append(dist.P2POp(dist.isend, t[lo:lo + r], slab.rank - 1, group))
        ops.append(dist.P2POp(dist.irecv, t[lo - r:lo], slab.rank - 1, group))
    if slab.r_hi:
        ops.append(dist.P2POp(dist.isend, t[hi - r:hi], slab.rank + 1, group))
        ops.append(dist.P2POp(dist.irecv, t[hi:hi + r], slab.rank + 1, group))
    return ops


class _HostStaged:
    """work handle of the rehearsal path below: wait() finishes the exchange and copies back"""

    def __init__(self, works, copies):
        self.works, self.copies = works, copies

    def wait(self):
        for w in self.works:
            w.wait()
        for dst, src in self.copies:
            dst.copy_(src)


def exchange_halos(slab: Slab, t: torch.Tensor, group=None, ops: Optional[List[dist.P2POp]] = None) -> list:
    """start the exchange; returns the work handles (call .wait() on each).  `ops`: the result of an earlier
    halo_ops(slab, t, group) for the same buffer (a time loop reuses two buffers: build the descriptors once).

    Production: RCCL send/recv straight between device buffers (backend "nccl").  If the process
    group is gloo and the buffer lives on a GPU (multi-rank rehearsal on a one-GPU box, where RCCL
    cannot pair two ranks on one device) the planes are staged through host memory instead: same
    geometry, same ordering, different transport."""
    if ops is None:
        ops = halo_ops(slab, t, group)
    if not ops:
        return []
    if t.is_cuda and dist.get_backend(group) == "gloo":
        staged, copies = [], []
        torch.cuda.current_stream().synchronize()
        for op in ops:
            host = op.tensor.detach().cpu() if op.op is dist.isend else torch.empty_like(op.tensor, device="cpu")
            staged.append(dist.P2POp(op.op, host, op.peer, group))
            if op.op is dist.irecv:
                copies.append((op.tensor, host))
        return [_HostStaged(dist.batch_isend_irecv(staged), copies)]
    return dist.batch_isend_irecv(ops)


class SlabComm:
    """The C-ABI communicator of include/neptune_hip.h section 8: libneptune_hip.so itself moves the halo planes (no
    Python, no torch.distributed in the step).  transport = "rccl": grouped ncclSend / ncclRecv; "peer": every rank
    pushes its edge planes into its neighbour's ghost planes with peer copies (hipIpc mappings, SDMA over xGMI, no CU
    moves data; the ranks of one node, and the one transport two processes on ONE device can use).  One per process;
    collective to create.  `peer_lo` / `peer_hi` default to rank -/+ 1.

    Both transports issue work on a stream of their own: synchronise the compute stream (or wait for the step) before
    issuing torch.distributed collectives on the same device -- two communicators progressing concurrently can deadlock."""

    TRANSPORTS = {"rccl": 0, "peer": 1}

    def __init__(self, rank: int, world: int, unique_id: Optional[bytes] = None, transport: str = "rccl"):
        import ctypes as C
        from . import _capi
        self._capi, self._C = _capi, C
        self.lib = _capi.load()
        self.rank, self.world, self.transport = rank, world, transport
        buf = C.create_string_buffer(unique_id, _capi.SLAB_ID_BYTES) if unique_id is not None else None
        self.ptr = self.lib.neptune_hip_slab_comm_create_ex(self.TRANSPORTS[transport], buf, rank, world)
        if not self.ptr:
            raise RuntimeError(f"neptune_hip_slab_comm_create_ex({transport}): " + self.last_error())

    def status(self) -> None:
        """raises once a device-side wait of the peer transport has timed out (a neighbour died or fell behind by more
        than NEPTUNE_HIP_PEER_TIMEOUT_S): every exchange since then left stale ghost planes"""
        if self.lib.neptune_hip_slab_comm_status(self.ptr) != 0:
            raise RuntimeError("neptune_hip_slab_comm_status: " + self.last_error())

    def last_error(self) -> str:
        return (self.lib.neptune_hip_slab_last_error() or b"").decode()

    @classmethod
    def from_process_group(cls, group=None, device: Optional[torch.device] = None, transport: str = "rccl") -> "SlabComm":
        """rank 0 draws the unique id and broadcasts it over the torch.distributed group (any backend);
        every rank then joins the communicator on its current device"""
        from . import _capi
        import ctypes as C
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        lib = _capi.load()
        idbuf = C.create_string_buffer(_capi.SLAB_ID_BYTES)
        ok = 1
        if rank == 0:
            ok = 1 if lib.neptune_hip_slab_unique_id_ex(cls.TRANSPORTS[transport], idbuf) == 0 else 0
        dev = device if device is not None else (torch.device("cuda") if dist.get_backend(group) == "nccl" else torch.device("cpu"))
        t = torch.tensor(list(idbuf.raw) + [ok], dtype=torch.uint8, device=dev)
        if world > 1:
            dist.broadcast(t, dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        raw = bytes(t.cpu().tolist())
        if raw[-1] != 1:
            raise RuntimeError("neptune_hip_slab_unique_id failed on rank 0: " + (lib.neptune_hip_slab_last_error() or b"").decode())
        return cls(rank, world, raw[:-1], transport)

    def exchange(self, slab: "Slab", t: torch.Tensor, stream: Optional[int] = None,
                 peer_lo: Optional[int] = None, peer_hi: Optional[int] = None) -> None:
        """refresh the ghost planes of the dense local buffer `t` on `stream` (asynchronous)"""
        if tuple(t.shape) != slab.local_shape or not t.is_contiguous() or not t.is_cuda:
            raise ValueError("tensor is not this slab's dense local device buffer")
        plane_bytes = t[0].numel() * t.element_size()
        st = int(torch.cuda.current_stream().cuda_stream) if stream is None else stream
        rc = self.lib.neptune_hip_halo_exchange(self.ptr, t.data_ptr(), plane_bytes, slab.n_own, slab.r_lo, slab.r_hi,
                                                slab.rank - 1 if peer_lo is None else peer_lo,
                                                slab.rank + 1 if peer_hi is None else peer_hi, st)
        if rc != 0:
            raise RuntimeError(f"neptune_hip_halo_exchange: {self._capi.ERROR_NAMES.get(rc, rc)}: {self.last_error()}")

    def close(self) -> None:
        if getattr(self, "ptr", None):
            self.lib.neptune_hip_slab_comm_destroy(self.ptr)
            self.ptr = None


RcclComm = SlabComm   # the name of the round-2 API


class ShardedApply:
    """One apply over a slab-decomposed field, exchange overlapped with the interior.  `body`: a built-in
    body id (neptune_hip._capi.BODY_*) or the geometry-level entry of a lowered module's apply
    (LoweredModule.geom_entry) -- any user stencil; `slab.radius` must be its reach along dim 0
    (GeomEntry.halo0).  Several inputs: pass lists of fields; every input's ghost planes are exchanged.

    Transport: `comm` = an RcclComm -> the whole step (exchange on a communication stream, interior, edge planes)
    is ONE call into libneptune_hip.so (neptune_hip_slab_apply); without it the exchange goes through
    torch.distributed point-to-point ops on `group` (RCCL for the default nccl group; a gloo group = staged through
    host memory: the rehearsal path of the tests)."""

    def __init__(self, slab: Slab, body: int, bounds: Box, cfg=None, overlap: bool = True, group=None,
                 comm: Optional[RcclComm] = None, peers: Optional[Tuple[int, int]] = None):
        from . import apply as _apply  # GPU path only
        self.group = group        # process group of the halo exchange (None = default; a gloo group = host-staged)
        self.rccl = comm
        self.peers = peers if peers is not None else (slab.rank - 1, slab.rank + 1)
        self._plans = {}
        self._apply = _apply
        self.slab = slab
        self.body = body
        self.bounds = slab.clip_bounds(bounds)
        self.cfg = cfg
        self.overlap = overlap
        self.interior, self.edges = slab.regions()
        self.compute = torch.cuda.current_stream()
        # the stream of the torch.distributed route (RCCL): default priority unless NEPTUNE_HIP_COMM_PRIORITY=high -- the C
        # plan's rule for the RCCL transport (csrc/runtime/slab_rccl.hip: copy kernels at the greatest priority take the
        # interior's CUs on one GPU; between devices the head start may pay: bench.py measures both)
        import os
        self.comm = torch.cuda.Stream(priority=-1 if os.environ.get("NEPTUNE_HIP_COMM_PRIORITY") == "high" else 0)
        self.ready = torch.cuda.Event()
        self.halo_done = torch.cuda.Event()
        self._cache = {}

    # ---- launch records: geometry structs and argument arrays are built once per (input, output)
    # pair; a step is then a handful of ctypes calls (keeps the host ahead of sub-millisecond kernels)
    def _records(self, fin, fout):
        fins = list(fin) if isinstance(fin, (list, tuple)) else [fin]
        key = (tuple(f.ptr for f in fins), fout.ptr)
        rec = self._cache.get(key)
        if rec is None:
            import ctypes as C
            lib = self._apply._capi.load()
            mk = lambda region: (self._apply.geom_for(fins, fout, self.bounds, region), self._apply._in_array(fins))
            rec = {
                "halo_ops": None,      # built at the first exchange (needs the process group; region-only users never exchange)
                "tensors": [f.tensor for f in fins],
                "whole": mk(self._own_region()),
                "interior": mk(self.interior) if self.interior is not None else None,
                "edges": [mk(r) for r in self.edges],
                "fn": lib.neptune_hip_apply_builtin,
                "cfg": C.byref(self.cfg) if self.cfg is not None else None,
                "byref": C.byref,
            }
            self._cache[key] = rec
        return rec

    def _launch(self, rec, which, fout, st) -> None:
        g, ins = which
        if hasattr(self.body, "fn"):   # a lowered module's apply
            rc = self.body.fn(rec["byref"](g), ins, fout.ptr, st, rec["cfg"])
        else:
            rc = rec["fn"](self.body, rec["byref"](g), ins, fout.ptr, st, rec["cfg"])
        if rc < 0:
            raise self._apply._capi.NeptuneHipError(rc, getattr(self.body, "symbol", "neptune_hip_apply_builtin"))

    def _cplan(self, fins, fout):
        """the C-side plan for this number of inputs / element type (geometry only: independent of the buffers)"""
        key = (len(fins), fout.dtype)
        plan = self._plans.get(key)
        if plan is None:
            import ctypes as C
            capi = self._apply._capi
            lib = capi.load()
            g = self._apply.geom_for(fins, fout, self.bounds)
            is_entry = hasattr(self.body, "fn")
            fn = C.cast(self.body.fn, C.c_void_p) if is_entry else None
            plan = lib.neptune_hip_slab_plan_create(self.rccl.ptr, fn, -1 if is_entry else self.body, fout.dtype, C.byref(g),
                                                    self.slab.radius, self.slab.r_lo, self.slab.r_hi, self.peers[0], self.peers[1],
                                                    C.byref(self.cfg) if self.cfg is not None else None)
            if not plan:
                raise RuntimeError("neptune_hip_slab_plan_create: " + self.rccl.last_error())
            self._plans[key] = plan
        return plan

    def __del__(self):
        try:
            lib = self._apply._capi.load()
            for plan in self._plans.values():
                lib.neptune_hip_slab_plan_destroy(plan)
        except Exception:   # interpreter shutdown
            pass

    def timing(self, on: bool) -> None:
        """C-ABI transport only: make the following steps record where their time goes (neptune_hip_slab_plan_timing)"""
        lib = self._apply._capi.load()
        for plan in self._plans.values():
            lib.neptune_hip_slab_plan_timing(plan, 1 if on else 0)

    def read_timing(self) -> Optional[dict]:
        """averages over the steps since timing(True): exchange / interior / edge wait / edges / step, milliseconds"""
        import ctypes as C
        lib = self._apply._capi.load()
        for plan in self._plans.values():
            out = (C.c_double * 6)()
            if lib.neptune_hip_slab_plan_timing_read(plan, out) == 0 and out[5] > 0:
                return {"exchange_ms": out[0], "interior_ms": out[1], "edge_wait_ms": out[2], "edges_ms": out[3],
                        "step_ms": out[4], "steps": int(out[5])}
        return None

    def __call__(self, fin, fout) -> None:
        slab = self.slab
        st = int(self.compute.cuda_stream)
        if self.rccl is not None:
            fins = list(fin) if isinstance(fin, (list, tuple)) else [fin]
            rc = self._apply._capi.load().neptune_hip_slab_apply(self._cplan(fins, fout), self._apply._in_array(fins), fout.ptr, st,
                                                                 1 if self.overlap else 0)
            if rc != 0:
                raise RuntimeError(f"neptune_hip_slab_apply: {self._apply._capi.ERROR_NAMES.get(rc, rc)}: {self.rccl.last_error()}")
            return
        rec = self._records(fin, fout)
        if slab.world == 1:
            self._launch(rec, rec["whole"], fout, st)
            return
        # 1. exchange the input's edge planes on the comm stream, once the input is complete
        self.ready.record(self.compute)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(self.ready)
            if rec["halo_ops"] is None:
                rec["halo_ops"] = [op for t in rec["tensors"] for op in halo_ops(slab, t, self.group)]
            works = exchange_halos(slab, rec["tensors"][0], self.group, rec["halo_ops"])
            for w in works:
                w.wait()              # stream-ordered: the comm stream waits for RCCL, the host does not
            self.halo_done.record(self.comm)
        if not self.overlap:
            self.compute.wait_event(self.halo_done)
        # 2. interior planes overlap the exchange
        if rec["interior"] is not None:
            self._launch(rec, rec["interior"], fout, st)
        # 3. edge planes once the ghosts have landed
        self.compute.wait_event(self.halo_done)
        for e in rec["edges"]:
            self._launch(rec, e, fout, st)

    def _own_region(self) -> Box:
        lo, hi = self.slab.owned_planes()
        shape = self.slab.local_shape
        return [lo] + [0] * (len(shape) - 1), [hi] + list(shape[1:])


class ShardedModule:
    """A lowered module (neptune_hip.lowering.LoweredModule) called on a slab-decomposed field, one
    process per GPU.  The module is compiled once for the GLOBAL boxes its types declare; under
    neptune_hip_set_slab() its functions read their memref arguments as this rank's local buffers and
    clip every apply / store / reduce to the owned planes (include/neptune_hip.h, lowered_runtime.hpp).

        sm = ShardedModule(mod, slab)
        sm.call("entry", out_local, in_local)          # exchanges in_local's ghost planes, then runs
        total = sm.call("norm2", u_local)              # scalar results: partial sums, all-reduced

    `slab.radius` must cover the reach of the module's applies along dim 0 (an apply that reaches
    beyond the ghost planes it is given aborts: it would read outside its input).  A function whose
    applies are chained through a stencil (an apply reading neighbouring planes of a value computed in
    the same call) cannot run sharded -- it needs an exchange in the middle -- and aborts with a message
    saying so.

    Schedule (same as ShardedApply): the exchange runs on a communication stream and its completion event is handed to
    the runtime (neptune_hip_set_slab_pending); the function's first stencil apply launches its interior planes, waits
    for the event, then launches the planes next to the ghosts.  `comm` = an RcclComm selects the C-ABI transport;
    overlap=False (or a gloo group, whose planes are staged through the host) exchanges first and synchronises."""

    def __init__(self, module, slab: Slab, group=None, comm: Optional[RcclComm] = None, overlap: bool = True,
                 peers: Optional[Tuple[int, int]] = None):
        from . import _capi
        self.module = module
        self.slab = slab
        self.group = group
        self.rccl = comm            # C-ABI transport (RcclComm); None = torch.distributed point-to-point on `group`
        self.overlap = overlap
        self.peers = peers if peers is not None else (slab.rank - 1, slab.rank + 1)
        self._lib = _capi.load()
        self._comm_stream = None    # created at the first overlapped call (needs a device)
        self._ready = self._halo_done = self._ready0 = None

    def local_empty(self, dtype=torch.float64, device="cuda") -> torch.Tensor:
        return torch.empty(self.slab.local_shape, dtype=dtype, device=device)

    def _device_exchange(self, tensors) -> bool:
        """start the halo exchange of `tensors` on the communication stream and leave its completion event with the
        runtime (neptune_hip_set_slab_pending): the lowered function overlaps its first stencil apply's interior with
        it.  Returns False when this transport cannot run beside the compute stream (gloo staging)."""
        lib = self._lib
        staged = self.rccl is None and dist.get_backend(self.group) == "gloo"
        if staged or not self.overlap or not all(t.is_cuda for t in tensors):
            return False
        if self._comm_stream is None:
            import os
            self._comm_stream = torch.cuda.Stream(priority=-1 if os.environ.get("NEPTUNE_HIP_COMM_PRIORITY") == "high" else 0)
            self._ready, self._halo_done = lib.neptune_hip_event_create(), lib.neptune_hip_event_create()
            self._ready0 = lib.neptune_hip_event_create()
        cur = int(torch.cuda.current_stream().cuda_stream)
        comm = int(self._comm_stream.cuda_stream)
        lib.neptune_hip_event_record(self._ready, cur)          # the inputs are complete on the caller's stream
        lib.neptune_hip_stream_wait_event(comm, self._ready)
        if cur != 0:
            lib.neptune_hip_stream_wait_event(None, self._ready)   # lowered functions launch on the null stream
            # ... and an earlier (asynchronous) call's kernels may still be writing planes there: the exchange must not
            # send or overwrite them before they are done (torch streams are non-blocking: no implicit ordering)
            lib.neptune_hip_event_record(self._ready0, None)
            lib.neptune_hip_stream_wait_event(comm, self._ready0)
        if self.rccl is not None:
            for t in tensors:
                self.rccl.exchange(self.slab, t, stream=comm, peer_lo=self.peers[0], peer_hi=self.peers[1])
        else:
            with torch.cuda.stream(self._comm_stream):
                for t in tensors:
                    for w in exchange_halos(self.slab, t, self.group):
                        w.wait()            # stream-ordered: the communication stream waits for RCCL, the host does not
        lib.neptune_hip_event_record(self._halo_done, comm)
        return True

    def call(self, name: str, *args, exchange: Optional[Sequence[int]] = None):
        """`exchange`: indices of the arguments whose ghost planes are refreshed before the call
        (default: every tensor argument but the first, the reference's @entry(out, in...) convention; pass
        () when the caller has already exchanged)"""
        slab = self.slab
        res = self.module.signatures[name]["result"]
        if slab.world > 1 and res and res["kind"] == "scalar" and res.get("scalar") == "derived":
            raise ValueError(f"@{name} returns a scalar computed from a neptune_ir.reduce result; on a slab decomposition that "
                             "result is only this rank's partial sum, so the function would be evaluated on it and the "
                             "ranks' values could not be combined.  Return the bare reduce (ShardedModule adds the ranks' sums) "
                             "and finish the arithmetic on the total.")
        if exchange is None:
            exchange = [i for i, a in enumerate(args) if i > 0 and hasattr(a, "shape")]
            if len(args) == 1:
                exchange = [0]
        pending = False
        if (slab.r_lo or slab.r_hi) and len(exchange):
            tensors = [getattr(args[i], "tensor", args[i]) for i in exchange]
            pending = self._device_exchange(tensors)
            if not pending:
                for t in tensors:
                    if self.rccl is not None and t.is_cuda:
                        self.rccl.exchange(slab, t, peer_lo=self.peers[0], peer_hi=self.peers[1])
                    else:
                        for w in exchange_halos(slab, t, self.group):
                            w.wait()
                if any(getattr(getattr(a, "tensor", a), "is_cuda", False) for a in args):
                    torch.cuda.current_stream().synchronize()   # lowered functions launch on the null stream
        rc = self._lib.neptune_hip_set_slab(slab.start, slab.stop, slab.r_lo, slab.r_hi)
        if rc != 0:
            raise ValueError("bad slab")
        if pending:   # exchange beside the interior of the function's first stencil apply
            self._lib.neptune_hip_set_slab_pending(self._halo_done)
        try:
            ret = self.module.call(name, *args)
        finally:
            self._lib.neptune_hip_clear_slab()
        sig = self.module.signatures[name]
        if sig["result"] and sig["result"]["kind"] == "scalar" and slab.world > 1 and sig["result"].get("scalar") != "uniform":
            # a bare reduce returned this rank's partial sum over its owned planes: the ranks' values add up.  (A scalar
            # computed FROM a reduce -- sqrt of it, a product of two -- never gets here: checked before the call.)
            t = torch.tensor([ret], dtype=torch.float64)
            if dist.get_backend(self.group) == "nccl":
                t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            ret = float(t.item())
        return ret

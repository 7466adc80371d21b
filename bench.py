#!/usr/bin/env python3
"""Headline benchmark of the NeptuneIR stencil hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W          (N > 1, one rank per GPU)

A "step" is one `neptune_ir.apply` of the 3-D 7-point Laplacian fixture
(tests/mlir_tests/conversion_tests/apply-3d-7pt.mlir's body) over the whole 1024^3 fp64 field
-- BASELINE.json's metric configuration -- ping-ponging between two device-resident fields, so
every step reads what the previous one wrote.  Inputs are resident in HBM before the timed
region; nothing crosses PCIe inside it.

N > 1: the field is cut into dim-0 slabs (strong scaling: the global problem stays 1024^3),
each step exchanges one halo plane per neighbour over RCCL/xGMI on a second stream, overlapped
with the interior update (neptune_hip/slab.py).

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for how every field is derived.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO / "neptune-pde-solver_amd"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s)

WORKLOADS = {
    # name: (body name, global shape, element bytes, stencil points)
    "3d7_1024": ("lap3d7_f64", (1024, 1024, 1024), 8, 7),
    "3d7_512": ("lap3d7_f64", (512, 512, 512), 8, 7),
    "2d5_8192": ("lap2d5_f64", (8192, 8192), 8, 5),
    "2d5_1024": ("lap2d5_f64", (1024, 1024), 8, 5),
    "3d27_512": ("lap3d27_f32", (512, 512, 512), 4, 27),
}
ORACLE_FN = {"lap3d7_f64": "lap3d7_f64", "lap2d5_f64": "lap2d5_f64", "lap3d27_f32": "lap3d27_f32"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="3d7_1024", choices=sorted(WORKLOADS))
    ap.add_argument("--kernel", default="auto", choices=["auto", "direct", "direct-flat", "march"])
    ap.add_argument("--variant", type=int, default=-1, help="march tile variant (-1 = library default)")
    ap.add_argument("--chunk", type=int, default=0, help="march planes per workgroup (0 = auto)")
    ap.add_argument("--no-autotune", action="store_true",
                    help="skip the plan-time tuning (untimed, before warm-up) and use the library's default tile")
    ap.add_argument("--no-overlap", action="store_true", help="N>1: exchange halos before the interior (debug)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="diagnostic: run the multi-rank code path with every rank on cuda:0 and gloo as transport "
                         "(halos staged through host memory); checks the control flow, not the speed")
    ap.add_argument("--emulate-rank", default="",
                    help="diagnostic, single process: 'R/W' runs the compute launches rank R of W would issue "
                         "(interior + edge regions of its slab, no exchange) to tune slab-sized kernels on one GPU")
    ap.add_argument("--allow-host-staging", action="store_true",
                    help="N>1: if no RCCL transport passes its start-up check, stage the halo planes through host memory over a "
                         "gloo group instead of failing (a PCIe number, not an xGMI one; reported in config.halo_transport)")
    ap.add_argument("--halo-transport", default="auto", choices=["auto", "c-abi", "torch"],
                    help="N>1: auto = the C-ABI exchange (libneptune_hip.so issues ncclSend/ncclRecv itself), falling back to "
                         "torch.distributed point-to-point ops if its start-up check fails; c-abi / torch force one")
    ap.add_argument("--fixed-input", action="store_true",
                    help="diagnostic: every step reads field 0 and writes field 1 (no ping-pong)")
    ap.add_argument("--builtin", action="store_true",
                    help="diagnostic: launch the runtime library's built-in copy of the fixture's body instead of the "
                         "module the lowering produces (default: fixture text -> libneptune_lowering emitter -> hipcc "
                         "-> the module's own geometry-level apply entry, i.e. the path north_star names)")
    ap.add_argument("--lowered", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--full-variants", action="store_true",
                    help="build the lowered module with every march tile of the library (NEPTUNE_HIP_FULL_VARIANTS=1: "
                         "a longer hipcc run) so that the plan-time tuning can choose among all of them")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-planes", type=int, default=0, help="dim-0 extent of the CPU sample (0 = auto)")
    ap.add_argument("--hbm-traffic-bytes", type=float, default=None,
                    help="per-launch HBM bytes from a separate rocprofv3 --pmc pass (reported as roofline.traffic)")
    return ap.parse_args()


def cpu_baseline(body_name, shape, elem_bytes, sample_planes):
    """Time the C oracle's faithful restatement of the reference lowering (1 thread, 3 passes,
    fresh malloc per apply) on a bounded dim-0 sample of the same workload, plus the fused
    all-core variant.  Test infrastructure used as the *baseline*, never as the product."""
    import numpy as np
    lib_path = REPO / "oracle" / "_build" / "liboracle.so"
    if not lib_path.exists():
        return None
    lib = C.CDLL(str(lib_path))
    nd = len(shape)
    ct = C.c_double if elem_bytes == 8 else C.c_float
    dt = np.float64 if elem_bytes == 8 else np.float32
    plane_cells = 1
    for n in shape[1:]:
        plane_cells *= n
    if sample_planes <= 0:
        # up to ~1.1e9 cells (the whole 1024^3 field: 3 x 8 GiB of host memory) keeps the CPU leg
        # (fill + three scalar 3-pass runs + five fused runs) around 15 s on the GPU node's host
        sample_planes = max(3, min(shape[0], int(1.1e9 // plane_cells)))
    sshape = (sample_planes,) + tuple(shape[1:])
    count = sample_planes * plane_cells
    fill = lib.ref_fill_hash_f64 if elem_bytes == 8 else lib.ref_fill_hash_f32
    fill.argtypes = [C.POINTER(ct), C.c_int64, C.c_int64, C.c_uint64]
    fill.restype = None
    u = np.empty(sshape, dt)
    out = np.zeros(sshape, dt)   # touch the destination like the reference driver does (out[i] = 0)
    fill(u.ctypes.data_as(C.POINTER(ct)), count, 0, 2024)
    lb = (C.c_int64 * nd)(*([1] * nd))
    ub = (C.c_int64 * nd)(*[n - 1 for n in sshape])
    updates = 1
    for n in sshape:
        updates *= (n - 2)
    res = {}
    for variant in ("entry", "fused"):
        fn = getattr(lib, f"ref_{variant}_{ORACLE_FN[body_name]}")
        fn.restype = C.c_int
        fn.argtypes = [C.POINTER(ct), C.POINTER(ct)] + [C.c_int64] * nd + [C.POINTER(C.c_int64)] * 2
        best = None
        reps = 3 if variant == "entry" else 5
        for _ in range(reps):
            t0 = time.perf_counter()
            rc = fn(out.ctypes.data_as(C.POINTER(ct)), u.ctypes.data_as(C.POINTER(ct)),
                    *[C.c_int64(n) for n in sshape], lb, ub)
            dtm = time.perf_counter() - t0
            assert rc == 0
            best = dtm if best is None else min(best, dtm)
        res[variant] = (updates / best, best)
    lib.ref_num_threads.restype = C.c_int
    threads = int(lib.ref_num_threads())
    dims = "x".join(str(n) for n in sshape)
    return {
        "value": res["entry"][0], "unit": "cell-updates/s", "cores": 1, "kind": "port",
        "sample": f"{dims} slab of the workload (same plane size), faithful restatement of the reference "
                  f"lowering: malloc + copy-through + scalar loop nest + store copy, best of 3 runs, {res['entry'][1]:.2f} s each",
        # the fused leg runs on OpenMP's default team = the CPUs this process may use (its affinity mask / cgroup
        # share), which on a shared GPU node is less than the machine's core count: both are reported
        "fused_all_cores": {"value": res["fused"][0], "cores": threads, "seconds": res["fused"][1],
                            "cores_note": "OpenMP default team: the CPUs in this process's affinity mask"},
        "host_cores": os.cpu_count(),
        "host_cores_usable": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
    }


def main():
    args = parse_args()
    # stdout carries exactly ONE JSON line.  RCCL prints a version banner and gloo its connection notes to file
    # descriptor 1 from C code: keep a private handle on the real stdout for the JSON line and point fd 1 at stderr
    # for everything else.
    json_out = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py: no HIP device visible; the NeptuneIR HIP backend has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)

    from neptune_hip import _capi, apply as nh_apply, fields, slab as slab_mod
    lib = _capi.load()          # ImportError if libneptune_hip.so is missing: fail loudly
    lib.neptune_hip_init(local_rank)

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    coll_dev = "cpu" if args.rehearse_on_one_gpu else "cuda"   # device of the few control tensors

    body_name, gshape, esize, points = WORKLOADS[args.workload]
    body = nh_apply.BODY_BY_NAME[body_name]
    builtin_body = body
    args.lowered = not args.builtin
    if args.lowered:
        if args.full_variants:
            os.environ["NEPTUNE_HIP_FULL_VARIANTS"] = "1"
        sys.path.insert(0, str(REPO / "tools"))
        import make_stencil_mlir
        from neptune_hip import lowering as nh_lowering
        kind = {"lap3d7_f64": "3d7", "lap2d5_f64": "2d5", "lap3d27_f32": "3d27"}[body_name]
        text = make_stencil_mlir.stencil_module(kind, list(gshape))
        if world > 1:                      # one rank fills the module cache, the others load from it
            if rank == 0:
                nh_lowering.compile_module(text)
            dist.barrier()
        module = nh_lowering.compile_module(text)
        body = module.geom_entry(make_stencil_mlir.KINDS[kind][2])     # the fixture's opdef (@lap3d, ...)
    dtype = nh_apply.BODY_DTYPE[builtin_body]
    rank_nd = len(gshape)
    gbox = ([0] * rank_nd, list(gshape))
    gbounds = ([1] * rank_nd, [n - 1 for n in gshape])
    kernel = {"auto": _capi.KERNEL_AUTO, "direct": _capi.KERNEL_DIRECT, "direct-flat": _capi.KERNEL_DIRECT,
              "march": _capi.KERNEL_MARCH}[args.kernel]
    cfg = nh_apply.make_cfg(kernel, args.variant, args.chunk, _capi.FLAG_DIRECT_FLAT if args.kernel == "direct-flat" else 0)

    emu = None
    if args.emulate_rank:
        if world != 1:
            sys.exit("--emulate-rank is a single-process diagnostic")
        er, ew = (int(x) for x in args.emulate_rank.split("/"))
        emu = (er, ew)
    sl = slab_mod.decompose(gbox, 1, rank, world) if emu is None else slab_mod.decompose(gbox, 1, emu[0], emu[1])
    # two device-resident local fields (owned planes + ghost planes), ping-pong
    plane_cells = 1
    for n in gshape[1:]:
        plane_cells *= n
    bufs = [fields.DeviceField(sl.local_lb, sl.local_ub, dtype) for _ in range(2)]
    # global deterministic field: value depends on the GLOBAL linear index, so every rank count
    # works on the same data
    bufs[0].fill_hash(2024, index_offset=(sl.local_lb[0] - gbox[0][0]) * plane_cells)
    bufs[1].tensor.zero_()
    # Plan-time tuning, untimed and before any warm-up or timed step: the library times its tiles on
    # exactly this rank's dominant launch (the interior region of its slab) and keeps the fastest.
    # Every tile computes the same bits; explicit --variant/--chunk/--kernel switch it off.
    autotuned = None
    if not args.no_autotune and args.variant < 0 and args.chunk == 0 and args.kernel == "auto":
        probe = slab_mod.ShardedApply(sl, body, gbounds, cfg=None)
        region = probe.interior if (world > 1 or args.emulate_rank) and probe.interior is not None else probe._own_region()
        cfg, tuned_ms = nh_apply.autotune_builtin(body, [bufs[0]], bufs[1], probe.bounds, region=region)
        autotuned = {"variant": int(cfg.variant), "chunk": int(cfg.chunk), "ms": tuned_ms}
        if int(cfg.kernel) == _capi.KERNEL_AUTO:
            cfg = nh_apply.make_cfg(kernel, args.variant, args.chunk)
        bufs[1].tensor.zero_()
        del probe
    # Halo transport.  RCCL builds its point-to-point channels on first use (seconds): do that here, outside every timed
    # or counted step, and CHECK the result -- the ghost planes of the freshly filled field already hold the right global
    # values, so: keep a copy, poison them, exchange, compare.  Order of preference:
    #   1. the C-ABI path (include/neptune_hip.h section 8): libneptune_hip.so issues the grouped ncclSend/ncclRecv on
    #      its own communication stream; a sharded step is ONE call into the library;
    #   2. torch.distributed point-to-point ops on the default (nccl = RCCL) group;
    #   3. only with --allow-host-staging: planes staged through host memory over a gloo group.
    # A transport that fails or delivers wrong planes on ANY rank is dropped on ALL ranks (agreed through an all-reduce);
    # if none is left the run ends with a non-zero exit code instead of printing a number.
    halo_group, rccl_comm, transport = None, None, ("none" if world == 1 else None)

    def ranks_agree(ok: bool) -> bool:
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=coll_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag.item()))

    def exchange_is_correct(do_exchange, what) -> bool:
        t = bufs[0].tensor
        lo, hi = sl.owned_planes()
        ok = True
        try:
            want_lo, want_hi = t[:lo].clone(), t[hi:].clone()
            t[:lo].fill_(float("nan"))
            t[hi:].fill_(float("nan"))
            torch.cuda.synchronize()
            do_exchange(t)
            torch.cuda.synchronize()
            ok = bool(torch.equal(t[:lo], want_lo)) and bool(torch.equal(t[hi:], want_hi))
            if not ok:
                print(f"[bench] rank {rank}: {what}: ghost planes differ from the neighbours' planes", file=sys.stderr, flush=True)
        except Exception as e:                      # noqa: BLE001 - whatever the transport raises means "unusable"
            ok = False
            print(f"[bench] rank {rank}: {what} failed ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
        bufs[0].fill_hash(2024, index_offset=(sl.local_lb[0] - gbox[0][0]) * plane_cells)
        torch.cuda.synchronize()
        return ok

    if world > 1 and args.rehearse_on_one_gpu:
        transport = "gloo-host-staged (rehearsal)"
    elif world > 1 and emu is None:
        if args.halo_transport in ("auto", "c-abi"):
            ok = True
            try:
                rccl_comm = slab_mod.RcclComm.from_process_group()
            except Exception as e:                  # noqa: BLE001
                ok = False
                print(f"[bench] rank {rank}: C-ABI communicator: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            if ranks_agree(ok):
                ok = exchange_is_correct(lambda t: rccl_comm.exchange(sl, t), "C-ABI halo exchange")
            if ranks_agree(ok):
                transport = "rccl, C ABI (neptune_hip_slab_apply: ncclSend/ncclRecv issued by libneptune_hip.so)"
            else:
                rccl_comm = None
        if transport is None and args.halo_transport in ("auto", "torch"):
            def torch_p2p(t):
                for w in slab_mod.exchange_halos(sl, t):
                    w.wait()
            if ranks_agree(exchange_is_correct(torch_p2p, "torch.distributed point-to-point exchange")):
                transport = "rccl, torch.distributed point-to-point"
        if transport is None:
            if not args.allow_host_staging:
                if rank == 0:
                    print("[bench] no RCCL halo transport passed its start-up check (see the messages above); refusing to "
                          "fall back to host staging without --allow-host-staging", file=sys.stderr, flush=True)
                dist.barrier()
                dist.destroy_process_group()
                sys.exit(3)
            halo_group, transport = dist.new_group(backend="gloo"), "gloo-host-staged (no RCCL transport passed its check)"
    elif emu is not None and (sl.r_lo or sl.r_hi):
        # single-process emulation of rank R of W: the same C-ABI step with the rank itself as both neighbours (an RCCL
        # communicator of one rank, loop-back send/recv): every stream operation and launch of the real step, with the
        # halo planes copied on the device instead of crossing xGMI
        rccl_comm = slab_mod.RcclComm(0, 1)
        transport = "rccl, C ABI, loop-back to the same rank (emulation)"
    op = slab_mod.ShardedApply(sl, body, gbounds, cfg=cfg, overlap=not args.no_overlap, group=halo_group, comm=rccl_comm,
                               peers=(0, 0) if emu is not None else None)
    sharded = op
    stream_ptr = fields.current_stream_ptr()
    def barrier():
        if world > 1:
            dist.barrier()

    def step(s):
        if args.fixed_input:
            op(bufs[0], bufs[1])
        else:
            op(bufs[s % 2], bufs[(s + 1) % 2])

    # Clock ramp: the chip leaves its idle clocks only after some tens of milliseconds of load, which
    # is longer than the whole run of the small workloads (512^3: 0.2-0.4 ms per step).  Spend ~1 s (0.3 s was not
    # enough for the first process on a fresh box: the 27-point workload then read 0.23-0.29 ms instead of 0.19)
    # of untimed steps first, then restore the initial fields so the W warm-up steps and the K timed
    # steps start from the same data whatever the ramp did.
    # The stencil iterations are not contractive (27-point fp32: values grow ~52x per step), so the fields are
    # restored every 10 ramp steps as well: launches on overflowed inf/NaN data draw less power and run ~7 %
    # faster than on real data (kernel trace of the 27-point workload: 181 us against 197 us), which would make
    # the ramp unrepresentative and skew a profiler's per-kernel average.
    t_ramp = time.perf_counter()
    n_ramp = 0
    while True:
        for s in range(10):
            step(s)
        bufs[0].fill_hash(2024, index_offset=(sl.local_lb[0] - gbox[0][0]) * plane_cells)
        torch.cuda.synchronize()
        n_ramp += 10
        go = time.perf_counter() - t_ramp < 1.0 and n_ramp < 6000
        if world > 1:   # every rank must run the same number of steps (each step is an exchange): rank 0 decides
            flag = torch.tensor([1 if go else 0], dtype=torch.int32, device=coll_dev)
            dist.broadcast(flag, 0)
            go = bool(int(flag.item()))
        if not go:
            break
    bufs[0].fill_hash(2024, index_offset=(sl.local_lb[0] - gbox[0][0]) * plane_cells)
    bufs[1].tensor.zero_()
    torch.cuda.synchronize()
    barrier()

    for s in range(args.warmup):
        step(s)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()

    ev0, ev1 = lib.neptune_hip_event_create(), lib.neptune_hip_event_create()
    t0 = time.perf_counter()
    lib.neptune_hip_event_record(ev0, stream_ptr)
    for s in range(args.warmup, args.warmup + args.steps):
        step(s)
    t_enqueued = time.perf_counter() - t0      # host time to enqueue the K steps (the device is still running them)
    lib.neptune_hip_event_record(ev1, stream_ptr)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ev_ms = lib.neptune_hip_event_elapsed_ms(ev0, ev1)

    if world > 1:
        t = torch.tensor([elapsed, ev_ms], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, ev_ms = float(t[0]), float(t[1])

    # guard against a silently dead run: the result must be finite and must have changed
    probe = bufs[(args.warmup + args.steps) % 2].tensor
    lo, hi = sl.owned_planes()
    mid = probe[(lo + hi) // 2]
    assert bool(torch.isfinite(mid).all()), "non-finite values in the result"
    assert float(mid.abs().max()) > 0.0, "result is identically zero"

    if rank == 0:
        cells = 1
        updates = 1
        for n in gshape:
            cells *= n
            updates *= (n - 2)
        ms_per_step = elapsed * 1e3 / args.steps
        value = updates * args.steps / elapsed
        # roofline of the dominant kernel.  Algorithmic bytes per launch = 2 * N * sizeof(T): every
        # input cell read once, every result cell written once; one launch covers
        # this rank's owned cells.  Duration = HIP-event time on the launch stream / launches.
        own_cells = sl.n_own * plane_cells
        if world == 1:
            alg_bytes = 2.0 * own_cells * esize
            kern_ms = ev_ms / args.steps
        else:
            alg_bytes = 2.0 * own_cells * esize
            kern_ms = ev_ms / args.steps  # per step on this rank's compute stream (interior + edges + waits)
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        plan = nh_apply.plan_builtin(builtin_body, [bufs[0]], bufs[1], sharded.bounds, region=sharded._own_region(), cfg=cfg)
        import ctypes as C
        g_own = nh_apply.geom_for([bufs[0]], bufs[1], sharded.bounds, sharded.interior if (world > 1 and sharded.interior is not None)
                                  else sharded._own_region())
        vidx = lib.neptune_hip_apply_builtin_variant(builtin_body, C.byref(g_own), C.byref(cfg))   # the dominant launch's tile
        vname = lib.neptune_hip_march_variant_name(rank_nd, vidx).decode() if plan == _capi.KERNEL_MARCH else ""
        # HBM traffic cannot be counted live (PMC needs rocprofv3): report the per-launch bytes of
        # the matching kernel/shape from the committed separate-pass profile, or null
        traffic, traffic_src = args.hbm_traffic_bytes, "command line" if args.hbm_traffic_bytes else None
        tfile = REPO / "profiles" / "traffic.json"
        if traffic is None and world == 1 and tfile.exists():
            key = f"{args.workload}|{lib.neptune_hip_kernel_name(plan).decode()}|{vname}"
            ent = json.loads(tfile.read_text()).get("entries", {}).get(key)
            if ent:
                traffic, traffic_src = ent["traffic_bytes_per_launch"], ent["source"]
        out = {
            "metric": "stencil-cell-updates/s",
            "value": value,
            "unit": "cell-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64" if esize == 8 else "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{rank_nd}-D {points}-point Laplacian apply, "
                            f"{'x'.join(str(n) for n in gshape)} {'fp64' if esize == 8 else 'fp32'}, "
                            f"interior bounds, copy-through boundary, ping-pong fields",
                "fixture": {"lap3d7_f64": "apply-3d-7pt.mlir", "lap2d5_f64": "apply-2d-5pt.mlir",
                            "lap3d27_f32": "apply-3d-27pt.mlir"}[body_name],
                "emulated_rank": args.emulate_rank or None,
                "decomposition": f"dim-0 slabs x{world}, 1 halo plane/neighbour over RCCL"
                                 + ("" if args.no_overlap else ", overlapped with interior") if world > 1 else "single GPU",
                "kernel": lib.neptune_hip_kernel_name(plan).decode(),
                "variant": vname,
                "chunk": int(cfg.chunk),
                "autotuned": autotuned,
                "halo_transport": transport,
                "host_enqueue_us_per_step": t_enqueued * 1e6 / args.steps,
                "body": "lowered module (NeptuneIR text -> emitter -> hipcc)" if args.lowered else "library built-in (same statements)",
            },
            "hbm_GBps": achieved * world if world > 1 else achieved,
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms": kern_ms,
                "per": "launch (whole field)" if world == 1 else "step on rank 0 (interior + edge launches)",
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(body_name, gshape, esize, args.cpu_sample_planes)
        else:
            out["cpu_baseline"] = None
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()

    lib.neptune_hip_event_destroy(ev0)
    lib.neptune_hip_event_destroy(ev1)
    if world > 1:
        dist.barrier()
    del op, sharded                      # the C-side plan (its stream and events) before its communicator
    if rccl_comm is not None:
        torch.cuda.synchronize()
        rccl_comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

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
each step exchanges one halo plane per neighbour over xGMI on a second stream, overlapped
with the interior update (include/neptune_hip.h section 8; neptune_hip/slab.py).  Every halo
transport is CHECKED at start-up (ghost planes poisoned, exchanged, compared), the ones that
pass are timed for a few steps and the fastest is used; a start-up watchdog ends a run whose
communicator set-up never returns, naming the stage.

After the timed region the run verifies itself (sampled planes of one more step, bit for bit
against the CPU oracle) and, on one GPU, times the other single-GPU BASELINE configurations.

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for how every field is derived.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO / "neptune-pde-solver_amd"))
sys.path.insert(0, str(REPO / "tools"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s)

WORKLOADS = {
    # name: (body name, global shape, element bytes, stencil points)
    "3d7_1024": ("lap3d7_f64", (1024, 1024, 1024), 8, 7),
    "3d7_512": ("lap3d7_f64", (512, 512, 512), 8, 7),
    "2d5_8192": ("lap2d5_f64", (8192, 8192), 8, 5),
    "2d5_1024": ("lap2d5_f64", (1024, 1024), 8, 5),
    "3d27_512": ("lap3d27_f32", (512, 512, 512), 4, 27),
}
KIND = {"lap3d7_f64": "3d7", "lap2d5_f64": "2d5", "lap3d27_f32": "3d27"}
FIXTURE = {"lap3d7_f64": "apply-3d-7pt.mlir", "lap2d5_f64": "apply-2d-5pt.mlir", "lap3d27_f32": "apply-3d-27pt.mlir"}
ORACLE_FN = {"lap3d7_f64": "lap3d7_f64", "lap2d5_f64": "lap2d5_f64", "lap3d27_f32": "lap3d27_f32"}
# the other single-GPU configurations of BASELINE.json, timed after the headline on one GPU (config.configs)
EXTRA_CONFIGS = ["3d7_512", "2d5_8192", "3d27_512"]
SEED = 2024


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
                    help="diagnostic: run the multi-rank code path with every rank on cuda:0 and gloo as process group; the "
                         "halo planes move over the peer transport (which two processes on one device can use) or, with "
                         "--halo-transport torch, staged through host memory; checks the control flow, not the speed")
    ap.add_argument("--emulate-rank", default="",
                    help="diagnostic, single process: 'R/W' runs the compute launches rank R of W would issue "
                         "(interior + edge regions of its slab) with a loop-back exchange, to tune slab-sized kernels on one GPU")
    ap.add_argument("--allow-host-staging", action="store_true",
                    help="N>1: if no device transport passes its start-up check, stage the halo planes through host memory over a "
                         "gloo group instead of failing (a PCIe number, not an xGMI one; reported in config.halo_transport)")
    ap.add_argument("--halo-transport", default="auto", choices=["auto", "c-abi", "peer", "torch"],
                    help="N>1: auto = check the C-ABI RCCL exchange (libneptune_hip.so issues ncclSend/ncclRecv itself), the C-ABI "
                         "peer-copy exchange (hipIpc mappings + SDMA pushes, no CU moves data) and torch.distributed "
                         "point-to-point ops, time the ones that pass and keep the fastest; c-abi / peer / torch force one")
    ap.add_argument("--startup-timeout", type=float, default=120.0,
                    help="seconds any single start-up stage (process group, communicator, first exchange) may take before the "
                         "watchdog prints the stage and exits with code 4")
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
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="N=1, default workload: skip the other single-GPU BASELINE configurations (config.configs)")
    ap.add_argument("--no-verify", action="store_true", help="skip the self-check against the oracle after the timed region")
    ap.add_argument("--compile-only", action="store_true",
                    help="lower + hipcc the module(s) this command line needs into the cache and exit without touching the GPU "
                         "(run before profiling: a profiled run must be a pure cache hit)")
    ap.add_argument("--cpu-sample-planes", type=int, default=0, help="dim-0 extent of the CPU sample (0 = auto)")
    ap.add_argument("--hbm-traffic-bytes", type=float, default=None,
                    help="per-launch HBM bytes from a separate rocprofv3 --pmc pass (reported as roofline.traffic)")
    return ap.parse_args()


class Watchdog:
    """A run whose communicator set-up blocks for ever burns the driver's whole time limit and leaves no diagnosis.
    Every start-up stage announces itself here with a time limit; if it has not finished by then the watchdog thread
    prints which stage of which rank is stuck and ends the process with exit code 4 (exit -- never re-exec: this
    process has initialised the GPU)."""

    def __init__(self, rank):
        self.rank = rank
        self.lock = threading.Lock()
        self.name, self.deadline, self.t0 = None, None, None
        self.history = []
        t = threading.Thread(target=self._run, daemon=True)
        t.start()

    def stage(self, name, limit):
        with self.lock:
            self._close()
            self.name, self.t0, self.deadline = name, time.monotonic(), time.monotonic() + limit

    def done(self):
        with self.lock:
            self._close()

    def _close(self):
        if self.name is not None:
            self.history.append((self.name, round(time.monotonic() - self.t0, 2)))
        self.name = self.deadline = None

    def _run(self):
        while True:
            time.sleep(0.5)
            with self.lock:
                name, deadline, t0 = self.name, self.deadline, self.t0
            if name is not None and time.monotonic() > deadline:
                print(f"[bench] rank {self.rank}: WATCHDOG: stage '{name}' has not returned after {time.monotonic() - t0:.0f} s; "
                      f"stages completed before it: {self.history}.  Exiting with code 4.", file=sys.stderr, flush=True)
                os._exit(4)


def fixture_text(body_name, shape):
    import make_stencil_mlir
    return make_stencil_mlir.stencil_module(KIND[body_name], list(shape))


def cpu_baseline(body_name, shape, elem_bytes, sample_planes):
    """Time the C oracle's faithful restatement of the reference lowering (1 thread, 3 passes,
    fresh malloc per apply) on a bounded dim-0 sample of the same workload, plus the fused
    all-core variant.  Test infrastructure used as the *baseline*, never as the product."""
    import numpy as np
    lib_path = REPO / "oracle" / "_build" / "liboracle.so"
    if not lib_path.exists():
        return None
    # the fused leg runs one OpenMP thread per CPU this process may use, set explicitly: the OpenMP runtime is already
    # loaded (torch brings one) and keeps the default it read then -- 128 threads on boxes whose affinity mask holds
    # 256 CPUs (an OMP_NUM_THREADS of the box's environment, or its cgroup's CPU share)
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    env_threads = os.environ.get("OMP_NUM_THREADS")
    lib = C.CDLL(str(lib_path))
    lib.ref_num_threads.restype = C.c_int
    default_threads = int(lib.ref_num_threads())
    lib.ref_set_num_threads.argtypes = [C.c_int]
    lib.ref_set_num_threads.restype = None
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
    fill(u.ctypes.data_as(C.POINTER(ct)), count, 0, SEED)
    lb = (C.c_int64 * nd)(*([1] * nd))
    ub = (C.c_int64 * nd)(*[n - 1 for n in sshape])
    updates = 1
    for n in sshape:
        updates *= (n - 2)
    res = {}
    # the fused all-core leg at two team sizes: the OpenMP runtime's own default and one thread per usable CPU (a
    # memory-bound loop on an SMT host is often faster on one thread per core); the faster one is reported
    for variant, team in (("entry", default_threads), ("fused", default_threads), ("fused_all", usable)):
        fn = getattr(lib, f"ref_{'entry' if variant == 'entry' else 'fused'}_{ORACLE_FN[body_name]}")
        fn.restype = C.c_int
        fn.argtypes = [C.POINTER(ct), C.POINTER(ct)] + [C.c_int64] * nd + [C.POINTER(C.c_int64)] * 2
        if variant == "fused_all" and team == default_threads:
            res[variant] = res["fused"]
            continue
        lib.ref_set_num_threads(team)
        best = None
        reps = 3 if variant == "entry" else 5
        for _ in range(reps):
            t0 = time.perf_counter()
            rc = fn(out.ctypes.data_as(C.POINTER(ct)), u.ctypes.data_as(C.POINTER(ct)),
                    *[C.c_int64(n) for n in sshape], lb, ub)
            dtm = time.perf_counter() - t0
            assert rc == 0
            best = dtm if best is None else min(best, dtm)
        res[variant] = (updates / best, best, team)
    fused_best = max(res["fused"], res["fused_all"], key=lambda r: r[0])
    threads = fused_best[2]
    dims = "x".join(str(n) for n in sshape)
    return {
        "value": res["entry"][0], "unit": "cell-updates/s", "cores": 1, "kind": "port",
        "sample": f"{dims} slab of the workload (same plane size), faithful restatement of the reference "
                  f"lowering: malloc + copy-through + scalar loop nest + store copy, best of 3 runs, {res['entry'][1]:.2f} s each",
        "fused_all_cores": {"value": fused_best[0], "cores": threads, "seconds": fused_best[1],
                            "cores_note": f"the faster of two OpenMP team sizes: the runtime's default ({default_threads} threads: "
                                          f"{res['fused'][0]:.3g} updates/s) and one thread per CPU of this process's affinity mask "
                                          f"({usable}: {res['fused_all'][0]:.3g}); OMP_NUM_THREADS in the environment: {env_threads!r}"},
        "host_cores": os.cpu_count(),
        "host_cores_usable": usable,
    }


def main():
    args = parse_args()
    body_name, gshape, esize, points = WORKLOADS[args.workload]
    args.lowered = not args.builtin
    if args.full_variants:
        os.environ["NEPTUNE_HIP_FULL_VARIANTS"] = "1"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    extra = (world == 1 and args.workload == "3d7_1024" and not args.emulate_rank and not args.no_extra_configs
             and args.kernel == "auto" and args.variant < 0 and args.chunk == 0 and args.lowered)

    if args.compile_only:
        from neptune_hip import lowering as nh_lowering
        if args.lowered:
            nh_lowering.compile_module(fixture_text(body_name, gshape), load=False)
        if extra:
            for name in EXTRA_CONFIGS:
                nh_lowering.compile_module(fixture_text(WORKLOADS[name][0], WORKLOADS[name][1]), load=False)
        return

    # stdout carries exactly ONE JSON line.  RCCL prints a version banner and gloo its connection notes to file
    # descriptor 1 from C code: keep a private handle on the real stdout for the JSON line and point fd 1 at stderr
    # for everything else.
    json_out = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    dog = Watchdog(rank)
    # the modules of the other configurations compile on host threads while this process imports torch, builds the
    # headline module and tunes it (hipcc runs as a child process; the ctypes call releases the GIL); joined before the
    # clock ramp so that nothing competes with the timed region
    precompile = None
    if extra:
        from concurrent.futures import ThreadPoolExecutor
        from neptune_hip import lowering as _lw
        _pool = ThreadPoolExecutor(max_workers=len(EXTRA_CONFIGS))
        precompile = [_pool.submit(_lw.compile_module, fixture_text(WORKLOADS[n][0], WORKLOADS[n][1]), None, True, None, False)
                      for n in EXTRA_CONFIGS]
    dog.stage("import torch", 600)
    import torch
    import torch.distributed as dist
    dog.done()

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
        dog.stage("torch.distributed.init_process_group", args.startup_timeout)
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dog.done()
    coll_dev = "cpu" if args.rehearse_on_one_gpu else "cuda"   # device of the few control tensors

    def ranks_agree(ok: bool) -> bool:
        if world == 1:
            return ok
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=coll_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag.item()))

    def max_over_ranks(*vals):
        if world == 1:
            return [float(v) for v in vals]
        t = torch.tensor(list(vals), dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t]

    def barrier():
        if world > 1:
            dist.barrier()

    if world > 1:   # the first collective builds torch's own communicator: a stage of its own
        dog.stage("first torch.distributed collective (all_reduce)", args.startup_timeout)
        ranks_agree(True)
        torch.cuda.synchronize()
        dog.done()

    builtin_body = nh_apply.BODY_BY_NAME[body_name]
    body = builtin_body
    if args.lowered:
        from neptune_hip import lowering as nh_lowering
        import make_stencil_mlir
        text = fixture_text(body_name, gshape)
        dog.stage("lowering + hipcc of the module", 900)
        if world > 1:                      # one rank fills the module cache, the others load from it
            if rank == 0:
                nh_lowering.compile_module(text, load=False)
            dist.barrier()
        module = nh_lowering.compile_module(text)
        dog.done()
        body = module.geom_entry(make_stencil_mlir.KINDS[KIND[body_name]][2])     # the fixture's opdef (@lap3d, ...)
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
    index_offset = (sl.local_lb[0] - gbox[0][0]) * plane_cells

    def refill():
        # global deterministic field: value depends on the GLOBAL linear index, so every rank count works on the same data
        bufs[0].fill_hash(SEED, index_offset=index_offset)
        bufs[1].tensor.zero_()
        torch.cuda.synchronize()

    refill()
    # Plan-time tuning, untimed and before any warm-up or timed step: the library times its tiles on
    # exactly this rank's dominant launch (the interior region of its slab) and keeps the fastest.
    # Every tile computes the same bits; explicit --variant/--chunk/--kernel switch it off.
    autotuned = None
    if not args.no_autotune and args.variant < 0 and args.chunk == 0 and args.kernel == "auto":
        probe = slab_mod.ShardedApply(sl, body, gbounds, cfg=None)
        region = probe.interior if (world > 1 or args.emulate_rank) and probe.interior is not None else probe._own_region()
        # half a second of launches first: candidates timed on idle clocks look slower than the ones measured after them
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < 0.5:
            for _ in range(10):
                nh_apply.apply_builtin(body, [bufs[0]], bufs[1], probe.bounds, region=region)
            torch.cuda.synchronize()
        cfg, tuned_ms = nh_apply.autotune_builtin(body, [bufs[0]], bufs[1], probe.bounds, region=region)
        autotuned = {"variant": int(cfg.variant), "chunk": int(cfg.chunk), "ms": tuned_ms}
        if int(cfg.kernel) == _capi.KERNEL_AUTO:
            cfg = nh_apply.make_cfg(kernel, args.variant, args.chunk)
        bufs[1].tensor.zero_()
        del probe

    # ---- halo transport ------------------------------------------------------------------------------------------
    # Every candidate is CHECKED before it is trusted -- the ghost planes of the freshly filled field already hold the
    # right global values, so: keep a copy, poison them, exchange, compare -- and a transport that fails or delivers
    # wrong planes on ANY rank is dropped on ALL ranks (agreed through an all-reduce).  The ones that pass run a few
    # untimed-for-the-result steps each and the fastest is kept (config.halo_transports records every verdict):
    #   rccl-c  the C-ABI path (include/neptune_hip.h section 8): libneptune_hip.so issues the grouped ncclSend/ncclRecv
    #           on its own priority stream; a sharded step is ONE call into the library
    #   peer-c  the same C plan on the peer-copy transport: each rank pushes its edge planes into its neighbour's ghost
    #           planes through an IPC mapping (hipMemcpyAsync: SDMA over xGMI, no CU moves data), handshake kernels
    #   torch   torch.distributed point-to-point ops on the default (nccl = RCCL) group
    # Only with --allow-host-staging: planes staged through host memory over a gloo group.  If none is left the run
    # ends with a non-zero exit code instead of printing a number.
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
        refill()
        return ok

    def make_op(comm=None, group=None, peers=None):
        return slab_mod.ShardedApply(sl, body, gbounds, cfg=cfg, overlap=not args.no_overlap, group=group, comm=comm, peers=peers)

    def step_with(op, s):
        if args.fixed_input:
            op(bufs[0], bufs[1])
        else:
            op(bufs[s % 2], bufs[(s + 1) % 2])

    def time_steps(op, n, warm):
        """ms per step of `n` steps after `warm` untimed ones, barrier to barrier, max over ranks"""
        refill()
        for s in range(warm):
            step_with(op, s)
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for s in range(warm, warm + n):
            step_with(op, s)
        torch.cuda.synchronize()
        barrier()
        return max_over_ranks((time.perf_counter() - t0) * 1e3 / n)[0]

    DESCR = {
        "rccl-c": "rccl, C ABI (neptune_hip_slab_apply: ncclSend/ncclRecv issued by libneptune_hip.so)",
        "peer-c": "peer copies, C ABI (neptune_hip_slab_apply: hipMemcpyAsync into the neighbour's IPC-mapped ghost planes, "
                  "handshake kernels; no CU moves data)",
        "torch": "rccl, torch.distributed point-to-point",
    }
    halo_group, comms, ops, verdicts, transport, chosen = None, {}, {}, {}, ("none" if world == 1 else None), None
    if world > 1 and emu is None:
        # order: the peer transport first (every wait of it is bounded: a failure costs the candidate, not the run), then
        # torch's point-to-point ops (the communicator the first all_reduce already proved), then a second RCCL communicator
        # of our own (ncclCommInitRank beside torch's: the one stage that could only be ended by the watchdog)
        wanted = {"auto": ["peer-c", "torch", "rccl-c"], "c-abi": ["rccl-c"], "peer": ["peer-c"], "torch": ["torch"]}[args.halo_transport]
        if args.rehearse_on_one_gpu:       # two processes cannot share a device under RCCL; the peer transport can
            wanted = [w for w in wanted if w == "peer-c"]
        for name in wanted:
            ok, comm = True, None
            if name in ("rccl-c", "peer-c"):
                dog.stage(f"communicator for the '{name}' transport", args.startup_timeout)
                try:
                    comm = slab_mod.SlabComm.from_process_group(transport="rccl" if name == "rccl-c" else "peer")
                except Exception as e:                  # noqa: BLE001
                    ok = False
                    print(f"[bench] rank {rank}: {name} communicator: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
                ok = ranks_agree(ok)
                if ok:
                    dog.stage(f"first checked halo exchange on the '{name}' transport", args.startup_timeout)
                    ok = ranks_agree(exchange_is_correct(lambda t, c=comm: c.exchange(sl, t), f"{name} halo exchange"))
            else:
                dog.stage("first checked halo exchange on the 'torch' transport", args.startup_timeout)

                def torch_p2p(t):
                    for w in slab_mod.exchange_halos(sl, t):
                        w.wait()
                ok = ranks_agree(exchange_is_correct(torch_p2p, "torch.distributed point-to-point exchange"))
            verdicts[name] = {"ok": ok}
            if ok:
                # ... with the communication stream at the greatest and at the default priority: on one GPU (loop-back) RCCL's
                # copy kernels at the greatest priority slow the interior down more than they gain; between devices the
                # exchange is bound by the link and should start first.  Measured, not guessed.
                for prio in ("high", "normal"):
                    key = f"{name}/{prio}"
                    dog.stage(f"timing a few steps on the '{key}' transport", args.startup_timeout)
                    os.environ["NEPTUNE_HIP_COMM_PRIORITY"] = prio       # read when the plan / stream is created: the first step
                    op = make_op(comm=comm)
                    good = True
                    try:
                        verdicts[key] = {"ok": True, "ms_per_step": time_steps(op, 10, 3)}
                        if comm is not None:
                            comm.status()
                    except Exception as e:                  # noqa: BLE001
                        print(f"[bench] rank {rank}: {key} steps: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
                        good = False
                    if ranks_agree(good):
                        comms[key], ops[key] = comm, op
                    else:
                        verdicts[key] = {"ok": False, "error": "steps failed"}
                        del op
                if not any(k.startswith(name + "/") for k in ops) and comm is not None:
                    torch.cuda.synchronize()
                    comm.close()
            elif comm is not None:
                comm.close()
            dog.done()
        os.environ.pop("NEPTUNE_HIP_COMM_PRIORITY", None)
        usable = [k for k in ops]
        if usable:
            chosen = min(usable, key=lambda k: verdicts[k]["ms_per_step"])
            if world > 1:    # every rank holds the same max-over-ranks times, but agree explicitly
                pick = torch.tensor([usable.index(chosen)], dtype=torch.int32, device=coll_dev)
                dist.broadcast(pick, 0)
                chosen = usable[int(pick.item())]
            transport = DESCR[chosen.split("/")[0]] + f"; communication stream priority: {chosen.split('/')[1]}"
            keep = comms[chosen]
            for k in usable:
                if k != chosen:
                    ops.pop(k)
            torch.cuda.synchronize()
            import gc
            gc.collect()
            for c in {id(c): c for k, c in comms.items() if c is not None and c is not keep}.values():
                c.close()
            comms = {chosen: keep}
        elif args.rehearse_on_one_gpu and args.halo_transport == "torch":
            transport = "gloo-host-staged (rehearsal)"
        elif not args.allow_host_staging:
            if rank == 0:
                print("[bench] no device halo transport passed its start-up check (see the messages above); refusing to "
                      "fall back to host staging without --allow-host-staging", file=sys.stderr, flush=True)
            dist.barrier()
            dist.destroy_process_group()
            sys.exit(3)
        else:
            halo_group, transport = dist.new_group(backend="gloo"), "gloo-host-staged (no device transport passed its check)"
    elif emu is not None and (sl.r_lo or sl.r_hi):
        # single-process emulation of rank R of W: the same C-ABI step with the rank itself as both neighbours (a
        # communicator of one rank, loop-back): every stream operation and launch of the real step, with the halo planes
        # copied on the device instead of crossing xGMI
        which = "peer" if args.halo_transport == "peer" else "rccl"
        comms["emu"] = slab_mod.SlabComm(0, 1, transport=which)
        transport = f"{which}, C ABI, loop-back to the same rank (emulation)"
    if chosen is not None:
        op = ops[chosen]
    elif emu is not None:
        op = make_op(comm=comms.get("emu"), peers=(0, 0))
    else:
        op = make_op(group=halo_group)
    sharded = op
    the_comm = comms.get(chosen) if chosen is not None else comms.get("emu")
    stream_ptr = fields.current_stream_ptr()

    def step(s):
        step_with(op, s)

    # Clock ramp: the chip leaves its idle clocks only after some tens of milliseconds of load, which
    # is longer than the whole run of the small workloads (512^3: 0.2-0.4 ms per step).  Spend ~1 s (0.3 s was not
    # enough for the first process on a fresh box: the 27-point workload then read 0.23-0.29 ms instead of 0.19)
    # of untimed steps first, then restore the initial fields so the W warm-up steps and the K timed
    # steps start from the same data whatever the ramp did.
    # The stencil iterations are not contractive (27-point fp32: values grow ~52x per step), so the fields are
    # restored every 10 ramp steps as well: launches on overflowed inf/NaN data draw less power and run ~7 %
    # faster than on real data (kernel trace of the 27-point workload: 181 us against 197 us), which would make
    # the ramp unrepresentative and skew a profiler's per-kernel average.
    if precompile is not None:
        dog.stage("hipcc of the other configurations' modules", 900)
        for f in precompile:
            f.result()
    dog.stage("clock ramp", 300)
    t_ramp = time.perf_counter()
    n_ramp = 0
    while True:
        for s in range(10):
            step(s)
        bufs[0].fill_hash(SEED, index_offset=index_offset)
        torch.cuda.synchronize()
        n_ramp += 10
        go = time.perf_counter() - t_ramp < 1.0 and n_ramp < 6000
        if world > 1:   # every rank must run the same number of steps (each step is an exchange): rank 0 decides
            flag = torch.tensor([1 if go else 0], dtype=torch.int32, device=coll_dev)
            dist.broadcast(flag, 0)
            go = bool(int(flag.item()))
        if not go:
            break
    refill()
    barrier()

    dog.stage("warm-up and timed steps", 600)
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
    dog.done()
    elapsed, ev_ms = max_over_ranks(elapsed, ev_ms)
    if the_comm is not None:
        the_comm.status()       # a device-side wait of the peer transport that timed out means stale ghost planes: not a result

    # guard against a silently dead run: the result must be finite and must have changed
    probe = bufs[(args.warmup + args.steps) % 2].tensor
    lo, hi = sl.owned_planes()
    mid = probe[(lo + hi) // 2]
    assert bool(torch.isfinite(mid).all()), "non-finite values in the result"
    assert float(mid.abs().max()) > 0.0, "result is identically zero"

    # ---- where a sharded step's time goes (untimed-for-the-result extra steps; C-ABI transports only) -----------------
    breakdown = None
    if (world > 1 or emu is not None) and hasattr(op, "rccl") and op.rccl is not None:
        dog.stage("per-step timing breakdown", 300)
        op.timing(True)
        for s in range(20):
            step(s)
        torch.cuda.synchronize()
        mine = op.read_timing()
        op.timing(False)
        if mine is not None:
            keys = ["exchange_ms", "interior_ms", "edge_wait_ms", "edges_ms", "step_ms"]
            worst = max_over_ranks(*[mine[k] for k in keys])
            breakdown = {"rank0": {k: round(mine[k], 4) for k in keys} if rank == 0 else None,
                         "max_over_ranks": {k: round(v, 4) for k, v in zip(keys, worst)},
                         "steps_averaged": mine["steps"],
                         "note": "HIP events inside neptune_hip_slab_apply: exchange on the communication stream (input ready -> "
                                 "ghost planes landed), interior and edge launches on the compute stream; edge_wait = how long "
                                 "after the interior's end the exchange ended (0 = hidden behind it)"}
        barrier()
        dog.done()

    # ---- the configuration the dominant launch really used -----------------------------------------------------------
    launched = _capi.LaunchCfg()
    dom_region = sharded.interior if ((world > 1 or emu is not None) and sharded.interior is not None) else sharded._own_region()
    nh_apply.apply_builtin(body, [bufs[0]], bufs[1], sharded.bounds, region=dom_region, cfg=cfg)
    torch.cuda.synchronize()
    have_launch = bool(lib.neptune_hip_last_launch(C.byref(launched)))

    # ---- the run verifies itself: one more step from the freshly filled field, sampled planes against the oracle -------
    verified = None
    if not args.no_verify:
        dog.stage("self-check against the oracle", 600)
        verified = verify_against_oracle(args, torch, dist, world, rank, sl, bufs, refill, step, body_name, gshape, esize,
                                         plane_cells, int(launched.chunk) if have_launch else 0, ranks_agree)
        dog.done()
        if verified is None:
            if world > 1:
                dist.barrier()
                dist.destroy_process_group()
            sys.exit(5)

    # ---- a plain copy of the same field, same process, same clocks: the measured ceiling the roofline fraction sits under ----
    # ("also report vs the measured-copy ceiling"; outside the timed region, N = 1 only)
    copy_ceiling = None
    if world == 1 and emu is None and not args.no_verify:
        dog.stage("copy ceiling", 120)
        src_t, dst_t = bufs[0].tensor, bufs[1].tensor
        nbytes2 = 2.0 * src_t.numel() * src_t.element_size()
        dt_code = _capi.F64 if esize == 8 else _capi.F32

        def timed(fn, reps=10):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        cur = int(torch.cuda.current_stream().cuda_stream)
        cands = {"torch tensor.copy_ (hipMemcpyAsync device to device)": timed(lambda: dst_t.copy_(src_t)),
                 "neptune_hip_store_full (what an un-elided whole-field neptune_ir.store costs)":
                     timed(lambda: lib.neptune_hip_store_full(dt_code, src_t.data_ptr(), dst_t.data_ptr(), src_t.numel(), cur or None))}
        # the library's 16-byte-per-lane streaming copies (grid-stride; 1-8 loads in flight per lane; plain / non-temporal)
        for mode in range(lib.neptune_hip_copy_mode_count()):
            cands[f"neptune_hip_time_copy mode {mode} (16 B per lane streaming copy kernel)"] = \
                lib.neptune_hip_time_copy(dst_t.data_ptr(), src_t.data_ptr(), src_t.numel() * src_t.element_size(), cur or None, mode, 3, 10)
        how, ms_c = min(cands.items(), key=lambda kv: kv[1])
        copy_ceiling = {"GBps": nbytes2 / ms_c / 1e6, "ms": ms_c, "how": how + ", 10 launches after 3 warm-ups, this process, after the timed region",
                        "all_ms": {k.split(" (")[0]: round(v, 4) for k, v in cands.items()}}
        dog.done()

    # ---- the other single-GPU BASELINE configurations ------------------------------------------------------------------
    configs = None
    if extra:
        dog.stage("the other single-GPU configurations", 900)
        del op, sharded
        bufs.clear()
        torch.cuda.empty_cache()
        configs = []
        for name in EXTRA_CONFIGS:
            configs.append(run_config(name, nh_apply, fields, slab_mod, lib, _capi, torch, chain=1))
        configs.append(run_config("3d7_1024", nh_apply, fields, slab_mod, lib, _capi, torch, chain=3))
        dog.done()

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
        alg_bytes = 2.0 * own_cells * esize
        kern_ms = ev_ms / args.steps       # N>1: per step on this rank's compute stream (interior + edges + waits)
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        kname = lib.neptune_hip_kernel_name(int(launched.kernel)).decode() if have_launch else ""
        vname = (lib.neptune_hip_march_variant_name(rank_nd, int(launched.variant)).decode()
                 if have_launch and int(launched.kernel) == _capi.KERNEL_MARCH else "")
        # HBM traffic cannot be counted live (PMC needs rocprofv3): report the per-launch bytes of
        # the matching kernel/shape from the committed separate-pass profile, or null
        traffic, traffic_src = args.hbm_traffic_bytes, "command line" if args.hbm_traffic_bytes else None
        tfile = REPO / "profiles" / "traffic.json"
        if traffic is None and world == 1 and tfile.exists():
            ent = json.loads(tfile.read_text()).get("entries", {}).get(f"{args.workload}|{kname}|{vname}")
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
                "fixture": FIXTURE[body_name] + (" (the committed text regenerated at this size by tools/make_stencil_mlir.py; "
                                                 "tests/test_lowering.py pins generator == committed file)"),
                "emulated_rank": args.emulate_rank or None,
                "decomposition": (f"dim-0 slabs x{world}, 1 halo plane/neighbour" + ("" if args.no_overlap else ", exchange overlapped with interior")
                                  if world > 1 else "single GPU"),
                "kernel": kname,
                "variant": vname,
                "chunk": int(launched.chunk) if have_launch else None,
                "autotuned": autotuned,
                "halo_transport": transport,
                "halo_transports": verdicts or None,
                "step_breakdown_ms": breakdown,
                "host_enqueue_us_per_step": t_enqueued * 1e6 / args.steps,
                "body": "lowered module (NeptuneIR text -> emitter -> hipcc)" if args.lowered else "library built-in (same statements)",
                "verified": verified,
                "watchdog_stages_s": dog.history,
                "configs": configs,
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
                "copy_ceiling": copy_ceiling,
                "frac_of_copy": (achieved / copy_ceiling["GBps"]) if copy_ceiling else None,
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
    dog.stage("shutdown", 120)
    if world > 1:
        dist.barrier()
    op = sharded = None                  # the C-side plan (its stream and events) before its communicator
    ops.clear()
    import gc
    gc.collect()
    torch.cuda.synchronize()
    for c in comms.values():
        if c is not None:
            c.close()
    if world > 1:
        dist.destroy_process_group()
    dog.done()


def verify_against_oracle(args, torch, dist, world, rank, sl, bufs, refill, step, body_name, gshape, esize, plane_cells, chunk,
                          ranks_agree):
    """One more step of exactly the timed configuration from the freshly filled field; sampled planes of this rank's owned
    part (the global boundary, the slab cuts, chunk seams, the middle) compared bit for bit with the numpy oracle run on
    the three input planes each needs (regenerated on the host from the same hash).  The oracle is the checker here, never
    the thing measured.  Returns the report string, or None after printing the mismatch."""
    import numpy as np
    sys.path.insert(0, str(REPO / "tests"))
    import helpers
    kind = KIND[body_name]
    npdt = np.float64 if esize == 8 else np.float32
    refill()
    step(0)                                       # reads bufs[0], writes bufs[1] (also with --fixed-input)
    torch.cuda.synchronize()
    lo, hi = sl.owned_planes()
    n0 = gshape[0]
    want_planes = {sl.start, sl.start + 1, sl.stop - 2, sl.stop - 1, (sl.start + sl.stop) // 2}
    for c in (chunk, 2 * chunk):                  # chunk seams of the dominant launch, counted from its first plane
        first = sl.start + (1 if sl.r_lo else 0)
        if chunk and first + c < sl.stop:
            want_planes |= {first + c - 1, first + c}
    if args.emulate_rank:     # a loop-back exchange fills the ghosts with this rank's own planes: the planes next to a cut are not the global result
        want_planes -= {sl.start} if sl.r_lo else set()
        want_planes -= {sl.stop - 1} if sl.r_hi else set()
    planes = sorted(p for p in want_planes if sl.start <= p < sl.stop)
    row_shape = tuple(gshape[1:])
    bad = []
    for p in planes:                              # p: global plane index
        got = bufs[1].tensor[lo + (p - sl.start)].detach().cpu().numpy()
        if p == 0 or p == n0 - 1:                 # copy-through: the input plane itself
            want = helpers.hash_field(row_shape, npdt, SEED, index_offset=p * plane_cells)
        else:
            slab3 = helpers.hash_field((3,) + row_shape, npdt, SEED, index_offset=(p - 1) * plane_cells)
            want = helpers.oracle_entry(kind, slab3)[1]
        if not helpers.bits_equal(got, want):
            bad.append(p)
            print(f"[bench] rank {rank}: VERIFY: plane {p} differs from the oracle\n" + helpers.mismatch_report(got, want),
                  file=sys.stderr, flush=True)
    ok = ranks_agree(not bad)
    if not ok:
        return None
    return (f"{len(planes)} planes per rank bit-exact vs the oracle (one step of the timed configuration from the freshly filled "
            f"field; rank 0 checked global planes {planes})")


def run_config(name, nh_apply, fields, slab_mod, lib, _capi, torch, chain=1, steps=200):
    """One of BASELINE.json's other single-GPU configurations: the workload's fixture through the lowering (text ->
    emitter -> hipcc -> geometry-level entry), plan-time tuned like the headline, 1 s clock ramp on real data, `steps`
    ping-pong steps timed with HIP events on the launch stream.  chain = 3: the step loop at three applies per pass over
    HBM (neptune_hip_step_loop_chain; bit-identical to one apply per launch)."""
    from neptune_hip import lowering as nh_lowering
    import make_stencil_mlir
    body_name, shape, esize, points = WORKLOADS[name]
    module = nh_lowering.compile_module(fixture_text(body_name, shape))
    entry = module.geom_entry(make_stencil_mlir.KINDS[KIND[body_name]][2])
    dtype = nh_apply.BODY_DTYPE[nh_apply.BODY_BY_NAME[body_name]]
    nd = len(shape)
    bounds = ([1] * nd, [n - 1 for n in shape])
    a = fields.DeviceField([0] * nd, list(shape), dtype)
    b = fields.DeviceField([0] * nd, list(shape), dtype)
    a.fill_hash(SEED)
    b.tensor.zero_()
    cells = 1
    for n in shape:
        cells *= n
    alg_bytes = 2.0 * cells * esize
    st = fields.current_stream_ptr()
    ev0, ev1 = lib.neptune_hip_event_create(), lib.neptune_hip_event_create()
    launched = _capi.LaunchCfg()
    if chain == 1:
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < 0.5:      # clocks up before anything is timed
            for _ in range(10):
                nh_apply.apply_builtin(entry, [a], b, bounds)
            torch.cuda.synchronize()
        cfg, _ = nh_apply.autotune_builtin(entry, [a], b, bounds)
        if int(cfg.kernel) == _capi.KERNEL_AUTO:
            cfg = None

        def run(n):
            for s in range(n):
                nh_apply.apply_builtin(entry, [a if s % 2 == 0 else b], b if s % 2 == 0 else a, bounds, cfg=cfg)
    else:
        cfg = None

        def run(n):
            nh_apply.step_loop(entry, a, b, bounds, n)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:         # clock ramp on real data (restored every 12 steps)
        run(12)
        a.fill_hash(SEED)
        torch.cuda.synchronize()
    a.fill_hash(SEED)
    b.tensor.zero_()
    run(6)
    torch.cuda.synchronize()
    if chain == 1:
        lib.neptune_hip_last_launch(C.byref(launched))
    a.fill_hash(SEED)
    torch.cuda.synchronize()
    lib.neptune_hip_event_record(ev0, st)
    run(steps)
    lib.neptune_hip_event_record(ev1, st)
    torch.cuda.synchronize()
    ms = lib.neptune_hip_event_elapsed_ms(ev0, ev1) / steps
    lib.neptune_hip_event_destroy(ev0)
    lib.neptune_hip_event_destroy(ev1)
    gbps = alg_bytes / (ms * 1e-3) / 1e9
    res = {
        "workload": f"{nd}-D {points}-point Laplacian apply, {'x'.join(map(str, shape))} {'fp64' if esize == 8 else 'fp32'}"
                    + ("" if chain == 1 else f", step loop at {chain} applies per pass over HBM"),
        "name": name if chain == 1 else f"{name}_chain{chain}",
        "ms": ms,                                  # per step (= per apply)
        "hbm_GBps": gbps,                          # 2 N sizeof(T) per step / ms: one-apply-per-pass accounting
        "frac": gbps / HBM_PEAK_GBPS,
        "steps": steps,
        "body": "lowered module",
    }
    if chain == 1:
        res["variant"] = lib.neptune_hip_march_variant_name(nd, int(launched.variant)).decode() if int(launched.kernel) == _capi.KERNEL_MARCH else ""
        res["chunk"] = int(launched.chunk)
        # a plain copy of a field of this size, same process: the ceiling under this line (the faster of the library's two best
        # streaming-copy flavours, 20 launches each)
        nbytes = cells * esize
        copy_ms = min(lib.neptune_hip_time_copy(b.ptr, a.ptr, int(nbytes), st, mode, 3, 20) for mode in (3, 5))
        res["copy_GBps"] = 2.0 * nbytes / copy_ms / 1e6
        res["frac_of_copy"] = gbps / res["copy_GBps"]
    else:
        res["variant"], res["chunk"] = f"apply_march2 NS={chain}", None
        res["note"] = "frac counts 2 N sizeof(T) per STEP; the kernel moves ~2.35 field passes per three steps, so values above 1 are expected"
    del a, b
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    main()

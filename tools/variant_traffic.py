#!/usr/bin/env python3
"""HBM traffic of EVERY march tile of one bench workload, so that bench.py's roofline.traffic is known
whichever tile the plan-time tuner picks.

  run  : tools/variant_traffic.py run WORKLOAD          (the program put under rocprofv3 --pmc ...)
         launches each tile variant REPS times in index order (after one fill), default chunk
  parse: tools/variant_traffic.py parse WORKLOAD FETCH_DIR WRITE_DIR [--update profiles/traffic.json]
         maps counter rows to variants by dispatch order, applies the gfx950 correction
         (FETCH_SIZE counts half of a wide read stream; both counters are KiB), prints a table
"""
import csv
import glob
import json
import os
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "neptune-pde-solver_amd"))
sys.path.insert(0, str(REPO))
REPS = 2
ROUND = os.environ.get("NEPTUNE_PROFILE_ROUND", "r02")


def run(workload):
    import torch
    from bench import WORKLOADS
    from neptune_hip import _capi, apply as nh_apply, fields
    body_name, shape, esize, _ = WORKLOADS[workload]
    lib = _capi.load()
    lib.neptune_hip_init(0)
    body = getattr(_capi, "BODY_" + body_name.upper())
    dt = _capi.F64 if esize == 8 else _capi.F32
    rank = len(shape)
    a = fields.DeviceField.hashed(shape, dt, seed=1)
    b = fields.DeviceField.empty_like(a)
    bounds = ([1] * rank, [n - 1 for n in shape])
    for v in range(lib.neptune_hip_march_variant_count(rank)):
        cfg = nh_apply.make_cfg(kernel=_capi.KERNEL_MARCH, variant=v)
        for _ in range(REPS):
            nh_apply.apply_builtin(body, [a], b, bounds, cfg=cfg)
        torch.cuda.synchronize()
    print("launched", lib.neptune_hip_march_variant_count(rank), "variants x", REPS)


def rows(d, counter):
    out = []
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:          # gpurun merges successive runs into the same directory: the newest pass counts
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "neptune_apply_march" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                    out.append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    out.sort()
    return [v for _, v in out]


def parse(workload, fetch_dir, write_dir, update=None):
    import ctypes as C
    from bench import WORKLOADS
    _, shape, esize, _ = WORKLOADS[workload]
    rank = len(shape)
    lib = C.CDLL(str(REPO / "neptune-pde-solver_amd/lib/libneptune_hip.so"))
    lib.neptune_hip_march_variant_name.restype = C.c_char_p
    nvar = lib.neptune_hip_march_variant_count(rank)
    fetch, write = rows(fetch_dir, "FETCH_SIZE"), rows(write_dir, "WRITE_SIZE")
    if len(fetch) != nvar * REPS or len(write) != nvar * REPS:
        sys.exit(f"expected {nvar * REPS} march dispatches per pass, got {len(fetch)} / {len(write)}")
    cells = 1
    for n in shape:
        cells *= n
    alg = 2 * cells * esize
    entries = {}
    print(f"# {workload}: per-launch HBM traffic of every march tile (FETCH_SIZE x2 x1024 + WRITE_SIZE x1024), algorithmic = {alg} B")
    for v in range(nvar):
        name = lib.neptune_hip_march_variant_name(rank, v).decode()
        f = sum(fetch[v * REPS:(v + 1) * REPS]) / REPS
        w = sum(write[v * REPS:(v + 1) * REPS]) / REPS
        t = 2 * f * 1024 + w * 1024
        print(f"v{v:<3d} {name:34s} fetch_kib={f:12.0f} write_kib={w:12.0f} traffic={t:.4g} B  x{t / alg:.3f} of algorithmic")
        entries[f"{workload}|neptune_apply_march|{name}"] = {
            "fetch_size_kib_per_launch": f, "write_size_kib_per_launch": w, "traffic_bytes_per_launch": t,
            "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": t / alg,
            "source": f"profiles/{ROUND}_variant_traffic_{workload}.txt"}
    if update:
        p = Path(update)
        doc = json.loads(p.read_text())
        # every entry of this workload is replaced: an entry's `source` must hold exactly the number it quotes
        for k in [k for k in doc["entries"] if k.startswith(workload + "|")]:
            del doc["entries"][k]
        doc["entries"].update(entries)
        p.write_text(json.dumps(doc, indent=1) + "\n")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        upd = sys.argv[sys.argv.index("--update") + 1] if "--update" in sys.argv else None
        parse(sys.argv[2], sys.argv[3], sys.argv[4], upd)

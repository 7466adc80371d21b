#!/usr/bin/env python3
"""Time every march tile variant / chunk for the benchmark workloads on the visible GPU and print
achieved algorithmic GB/s next to a plain 16 B/lane copy (the measured HBM ceiling).

    python tools/sweep.py [--workloads 3d7_1024,2d5_8192,...] [--chunks 0,32,64,128] [--reps 10]
Results also go to gpurun_out/sweep.jsonl (one JSON object per line)."""
import argparse
import json
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "neptune-pde-solver_amd"))

WORKLOADS = {
    "3d7_1024": ("lap3d7_f64", (1024, 1024, 1024)),
    "3d7_512": ("lap3d7_f64", (512, 512, 512)),
    "2d5_8192": ("lap2d5_f64", (8192, 8192)),
    "3d27_512": ("lap3d27_f32", (512, 512, 512)),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="3d7_1024,3d7_512,2d5_8192,3d27_512")
    ap.add_argument("--chunks", default="0")
    ap.add_argument("--variants", default="all")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--out", default=str(REPO / "gpurun_out" / "sweep.jsonl"))
    args = ap.parse_args()
    import torch
    from neptune_hip import _capi, apply, fields
    lib = _capi.load()
    lib.neptune_hip_init(0)
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    fout = open(args.out, "a")
    chunks = [int(c) for c in args.chunks.split(",")]
    for wl in args.workloads.split(","):
        body_name, shape = WORKLOADS[wl]
        body = apply.BODY_BY_NAME[body_name]
        dtype = apply.BODY_DTYPE[body]
        a = fields.DeviceField.hashed(shape, dtype, seed=1)
        b = fields.DeviceField.empty_like(a)
        nbytes = 2.0 * a.nbytes
        bounds = ([1] * len(shape), [n - 1 for n in shape])
        for mode in range(lib.neptune_hip_copy_mode_count()):
            ms = lib.neptune_hip_time_copy(b.ptr, a.ptr, a.nbytes, fields.current_stream_ptr(), mode, 2, args.reps)
            rec = {"workload": wl, "kernel": f"copy16_mode{mode}", "ms": ms, "GBps": nbytes / ms / 1e6}
            print(f"{wl:10s} {rec['kernel']:28s} {ms:9.4f} ms {rec['GBps']:8.1f} GB/s", flush=True)
            fout.write(json.dumps(rec) + "\n")
        ms = apply.time_builtin(body, [a], b, bounds, apply.make_cfg(_capi.KERNEL_DIRECT), 1, max(2, args.reps // 3))
        rec = {"workload": wl, "kernel": "direct", "ms": ms, "GBps": nbytes / ms / 1e6}
        print(f"{wl:10s} {'direct':28s} {ms:9.4f} ms {rec['GBps']:8.1f} GB/s", flush=True)
        fout.write(json.dumps(rec) + "\n")
        nv = lib.neptune_hip_march_variant_count(len(shape))
        variants = range(nv) if args.variants == "all" else [int(v) for v in args.variants.split(",")]
        for v in variants:
            name = lib.neptune_hip_march_variant_name(len(shape), v).decode()
            for ch in chunks:
                ms = apply.time_builtin(body, [a], b, bounds, apply.make_cfg(_capi.KERNEL_MARCH, v, ch), 2, args.reps)
                rec = {"workload": wl, "kernel": "march", "variant": v, "name": name, "chunk": ch, "ms": ms,
                       "GBps": nbytes / ms / 1e6}
                print(f"{wl:10s} march {v:2d} {name:20s} chunk={ch:4d} {ms:9.4f} ms {rec['GBps']:8.1f} GB/s "
                      f"({100 * rec['GBps'] / 8000:5.1f}% of 8 TB/s)", flush=True)
                fout.write(json.dumps(rec) + "\n")
                fout.flush()
        del a, b
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

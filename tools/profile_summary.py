#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh output directory: per-kernel time stats and PMC sums."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


import re


def short(name):
    """kernel name without the boilerplate; march kernels get the library's own tile name (template arguments of
    Tile<RJ, WJ, WK, dpp, nt-store, PF, nt-load, lds-J, tile-form, late-J-halo, K-halo lead>)"""
    m = re.search(r"neptune_apply_march<.*?Tile<([^>]*)>", name)
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        f = lambda i: i < len(a) and a[i] == "true"
        kd = a[10] if len(a) > 10 else "1"
        if f(8):
            tag = f"tile_rj{a[0]}_wj{a[1]}_wk{a[2]}"
        else:
            tag = f"rj{a[0]}_wj{a[1]}_wk{a[2]}_pf{a[5]}" + ("_lds" if f(7) else "") + ("_jhl" if f(9) else "") + \
                  (f"_kd{kd}" if kd != "1" else "") + ("_ntl" if f(6) else "")
        tag += ("" if f(3) else "_shfl") + ("" if f(4) else "_plainst")
        return "neptune_apply_march[" + tag + "]"
    m = re.search(r"neptune_apply_planes?<.*?Tile<([^>]*)>", name)
    if m:   # plane-in-LDS kernels (apply_plane.hpp): only RJ, WJ, WK, PF of the tile mean anything there
        a = [x.strip() for x in m.group(1).split(",")]
        return ("neptune_apply_planes[" if "apply_planes<" in name else "neptune_apply_plane[") + f"pln_rj{a[0]}_wj{a[1]}_wk{a[2]}_pf{a[5]}]"
    if "neptune_apply_plane" in name:
        return "neptune_apply_plane" + ("s" if "apply_planes<" in name else "")
    m = re.search(r"neptune_apply_march2<.*?, (\d+), (\d+), \d+>", name)
    if m:
        return f"neptune_apply_march2[rows{m.group(1)}x{m.group(2)}]"
    for key in ("neptune_apply_direct", "neptune_apply_rows", "neptune_reduce_apply_vec", "neptune_reduce_apply", "neptune_fill_hash",
                "neptune_copy16", "neptune_store_box", "neptune_vec_update"):
        if key in name:
            return key
    return name[:60]


print("# kernel stats (rocprofv3 --kernel-trace --stats)")
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print(f"{short(row.get('Name', '')):50s} calls={row.get('Calls')} avg_ns={row.get('AverageNs')} "
                  f"min_ns={row.get('MinNs')} max_ns={row.get('MaxNs')} total_ns={row.get('TotalDurationNs')} pct={row.get('Percentage')}")
print("# per-dispatch durations of the apply kernel (kernel trace)")
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_trace.csv"), recursive=True):
    durs = []
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "neptune_apply" in row.get("Kernel_Name", ""):
                durs.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    if durs:
        durs.sort()
        print(f"n={len(durs)} min={durs[0]} median={durs[len(durs)//2]} max={durs[-1]} mean={sum(durs)/len(durs):.0f} ns")
print("# PMC counters, summed per kernel over dispatches (then / dispatches)")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row.get("Kernel_Name", ""))
                acc[k][row.get("Counter_Name")] += float(row.get("Counter_Value", 0))
                cnt[k].add(row.get("Dispatch_Id"))
    for k, counters in acc.items():
        n = max(1, len(cnt[k]))
        for c, v in counters.items():
            print(f"{os.path.basename(d):40s} {k:50s} {c:22s} total={v:.6g} per_dispatch={v / n:.6g} dispatches={n}")

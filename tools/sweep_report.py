#!/usr/bin/env python3
"""Summarise gpurun_out/sweep.jsonl: copy ceiling, best march variants per workload."""
import collections
import json
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/sweep.jsonl"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = [json.loads(l) for l in open(path)]
by = collections.defaultdict(list)
for r in rows:
    by[r["workload"]].append(r)
for wl, rs in by.items():
    print("==", wl)
    copies = [r for r in rs if r["kernel"].startswith("copy")]
    print("  copy:", " ".join(f"{r['kernel'][-5:]}:{r['GBps']:.0f}" for r in copies))
    print("  direct:", " ".join(f"{r['GBps']:.0f}" for r in rs if r["kernel"] == "direct"))
    ms = sorted([r for r in rs if r["kernel"] == "march"], key=lambda r: -r["GBps"])
    for r in ms[:top]:
        print(f"  v{r['variant']:2d} {r['name']:30s} chunk={r['chunk']:4d} {r['ms']:.4f} ms {r['GBps']:.0f} GB/s "
              f"{r['GBps'] / 80:.1f}%")
    best = {}
    for r in ms:
        best.setdefault(r["variant"], r)
    print("  best per variant:", " ".join(f"v{v}:{b['GBps']:.0f}@{b['chunk']}" for v, b in sorted(best.items())))

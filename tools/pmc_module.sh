#!/usr/bin/env bash
# One rocprofv3 counter pass over tools/time_module.py (a lowered module's @entry), averaged per kernel and dispatch.
# usage: tools/pmc_module.sh <tag> "<COUNTER ...>" <file.mlir> [time_module args...]   -> gpurun_out/pmcm_<tag>/summary.txt
# (launch configuration through NEPTUNE_HIP_VARIANT / _CHUNK / _KERNEL as usual)
set -euo pipefail
TAG=${1:?tag}; CTRS=${2:?counters}; shift 2 || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmcm_$TAG
mkdir -p "$OUT"
# fill the module cache OUTSIDE the profiler: the profiled run must not start hipcc (exec hops under the preloaded tool)
python3 "$ROOT/tools/time_module.py" "$@" --compile-only
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT/raw" -- python3 "$ROOT/tools/time_module.py" "$@" > "$OUT/time.json" 2> "$OUT/err.txt" || { tail -20 "$OUT/err.txt"; exit 1; }
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, os, sys, collections
out = sys.argv[1]
files = sorted(glob.glob(os.path.join(out, "raw", "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
with open(files[-1]) as fh:
    for row in csv.DictReader(fh):
        k = row["Kernel_Name"][:70]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
for k, c in acc.items():
    if "neptune_apply" not in k: continue
    print(k, "dispatches", len(n[k]))
    for name, v in sorted(c.items()):
        print(f"   {name:28s} {v / len(n[k]):.6g} per dispatch")
PY
rm -rf "$OUT/raw"

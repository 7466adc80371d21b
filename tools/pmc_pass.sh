#!/usr/bin/env bash
# One rocprofv3 counter pass over bench.py, summed per kernel.
# usage: tools/pmc_pass.sh <tag> "<COUNTER ...>" [bench args...]   -> gpurun_out/pmc_<tag>/summary.txt
set -euo pipefail
TAG=${1:?tag}; CTRS=${2:?counters}; shift 2 || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
# fill the module cache OUTSIDE the profiler: the profiled run must not start hipcc
python3 "$ROOT/bench.py" "$@" --compile-only
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT/raw" -- python3 "$ROOT/bench.py" "$@" > "$OUT/bench.json" 2> "$OUT/err.txt" || { tail -20 "$OUT/err.txt"; exit 1; }
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, os, sys, collections
out = sys.argv[1]
files = sorted(glob.glob(os.path.join(out, "raw", "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
with open(files[-1]) as fh:
    for row in csv.DictReader(fh):
        k = row["Kernel_Name"][:90]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
for k, c in acc.items():
    if "neptune" not in k: continue
    print(k, "dispatches", len(n[k]))
    for name, v in sorted(c.items()):
        print(f"   {name:28s} {v / len(n[k]):.6g} per dispatch")
PY
find "$OUT" -name "*agent_info*" -delete

#!/usr/bin/env bash
# rocprofv3 kernel-trace + stats of tools/time_module.py on one module, then a PMC pass for the HBM-side traffic.
# usage: tools/profile_module.sh <tag> <file.mlir>   -> gpurun_out/prof_<tag>/{kernel_stats.csv,summary.txt}
set -euo pipefail
TAG=${1:?tag}; MLIR=${2:?mlir}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
# fill the module cache OUTSIDE the profiler
python3 "$ROOT/tools/time_module.py" "$MLIR" --compile-only
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -- python3 "$ROOT/tools/time_module.py" "$MLIR" --reps 50 > "$OUT/time.json" 2> "$OUT/err.txt" || { tail -20 "$OUT/err.txt"; exit 1; }
cp "$(find "$OUT/raw" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
rm -rf "$OUT/raw"
# FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950 (TCC has 4 counter slots, they need 3 + 2)
"$ROOT/tools/pmc_module.sh" "${TAG}_fetch" "FETCH_SIZE" "$MLIR" --reps 10 > /dev/null
"$ROOT/tools/pmc_module.sh" "${TAG}_write" "WRITE_SIZE" "$MLIR" --reps 10 > /dev/null
{
  echo "# $TAG: tools/time_module.py $(basename "$MLIR") under timeout -k 10 300 rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE WRITE_SIZE"
  cat "$OUT/time.json" | python3 -c 'import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print("# wall: %.4f ms per call, %.1f GB/s algorithmic, shape %s" % (d["ms_per_call"], d["GBps"], d["shape"]))'
  python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if "neptune" in row["Name"]:
        print("%-100s calls %s avg %.1f us min %.1f max %.1f" % (row["Name"][:100], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
PY
  grep -h "FETCH_SIZE\|WRITE_SIZE" "$ROOT/gpurun_out/pmcm_${TAG}_fetch/summary.txt" "$ROOT/gpurun_out/pmcm_${TAG}_write/summary.txt" | sed 's/$/  (KiB; FETCH_SIZE reports half of the bytes of wide streaming reads on gfx950: x2, MI355X_MICROARCH.md)/' 
} > "$OUT/summary.txt"
cat "$OUT/summary.txt"
